"""CPU oracle (numpy / torch-CPU) of the evaluation arithmetic that follows the decoder — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(wavelet_monodepth_amd/evaluation.py) never does.

Restates, function by function:
  compute_errors                KITTI/evaluate_depth.py:50-68
  batch_post_process_disparity  KITTI/evaluate_depth.py:71-79
  kitti_image_metrics           KITTI/evaluate_depth.py:268-307 (inline loop body of evaluate())
  compute_errors_nyu            NYUv2/utils.py:85-98
  nyu_prediction_chain          NYUv2/utils.py:213-226,247-250 (rescale, replicate pad, 2x bilinear, clamp, Eigen crop)
Pinning: compute_errors, batch_post_process_disparity and compute_errors_nyu are checked against outputs of the
reference's own functions (tests/golden/eval_reference.npz, made by tests/golden/make_golden_eval.py which executes
those three function bodies out of /root/reference).  kitti_image_metrics is inline code of a script that needs cv2
(absent here): it is restated, with cv2.resize(INTER_LINEAR) replaced by its documented half-pixel / edge-clamp
formula — "parity unpinned" for that resize step only.
"""
import numpy as np

MIN_DEPTH, MAX_DEPTH = 1e-3, 80.0      # evaluate_depth.py:85-86
STEREO_SCALE_FACTOR = 5.4              # evaluate_depth.py:35


def compute_errors(gt, pred):
    thresh = np.maximum((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).mean()
    a2 = (thresh < 1.25 ** 2).mean()
    a3 = (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def batch_post_process_disparity(l_disp, r_disp):
    """r_disp is already flipped back, as at the reference's call site (:204)."""
    _, h, w = l_disp.shape
    m_disp = 0.5 * (l_disp + r_disp)
    l, _ = np.meshgrid(np.linspace(0, 1, w), np.linspace(0, 1, h))
    l_mask = (1.0 - np.clip(20 * (l - 0.05), 0, 1))[None, ...]
    r_mask = l_mask[:, :, ::-1]
    return r_mask * l_disp + l_mask * r_disp + (1.0 - l_mask - r_mask) * m_disp


def resize_bilinear_cv2(src, H, W):
    """cv2.resize(src, (W, H)) with the default INTER_LINEAR on a float32 image: sample at (dst + 0.5) * scale - 0.5,
    indices clamped to the image with the fractional weight zeroed at the clamp."""
    h, w = src.shape
    fy = (np.arange(H, dtype=np.float32) + np.float32(0.5)) * np.float32(h / H) - np.float32(0.5)
    fx = (np.arange(W, dtype=np.float32) + np.float32(0.5)) * np.float32(w / W) - np.float32(0.5)
    y0 = np.floor(fy).astype(np.int64)
    x0 = np.floor(fx).astype(np.int64)
    wy = (fy - y0).astype(np.float32)
    wx = (fx - x0).astype(np.float32)
    wy[y0 < 0] = 0
    y0[y0 < 0] = 0
    wx[x0 < 0] = 0
    x0[x0 < 0] = 0
    wy[y0 >= h - 1] = 0
    y0[y0 >= h - 1] = h - 1
    wx[x0 >= w - 1] = 0
    x0[x0 >= w - 1] = w - 1
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    top = src[y0][:, x0] * (1 - wx)[None, :] + src[y0][:, x1] * wx[None, :]
    bot = src[y1][:, x0] * (1 - wx)[None, :] + src[y1][:, x1] * wx[None, :]
    return (top * (1 - wy)[:, None] + bot * wy[:, None]).astype(np.float32)


def kitti_image_metrics(pred_disp, gt_depth, eigen=True, pred_depth_scale_factor=1.0, disable_median_scaling=False):
    """-> (7 metrics, ratio or None, n_valid) for one image, evaluate_depth.py:272-307."""
    gt_height, gt_width = gt_depth.shape[:2]
    pred_disp = resize_bilinear_cv2(pred_disp, gt_height, gt_width)
    pred_depth = 1 / pred_disp
    if eigen:
        mask = np.logical_and(gt_depth > MIN_DEPTH, gt_depth < MAX_DEPTH)
        crop = np.array([0.40810811 * gt_height, 0.99189189 * gt_height,
                         0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)
        crop_mask = np.zeros(mask.shape)
        crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
        mask = np.logical_and(mask, crop_mask)
    else:
        mask = gt_depth > 0
    pred_depth = pred_depth[mask]
    gt = gt_depth[mask]
    pred_depth = pred_depth * np.float32(pred_depth_scale_factor)
    ratio = None
    if not disable_median_scaling:
        ratio = np.median(gt) / np.median(pred_depth)
        pred_depth = pred_depth * ratio
    pred_depth[pred_depth < MIN_DEPTH] = MIN_DEPTH
    pred_depth[pred_depth > MAX_DEPTH] = MAX_DEPTH
    return compute_errors(gt, pred_depth), ratio, int(mask.sum())


def compute_errors_nyu(pred, gt):
    """numpy form of NYUv2/utils.py:85-98 (torch there)."""
    y, x = gt, pred
    thresh = np.maximum(y / x, x / y)
    a1 = (thresh < 1.25).astype(np.float32).mean()
    a2 = (thresh < 1.25 ** 2).astype(np.float32).mean()
    a3 = (thresh < 1.25 ** 3).astype(np.float32).mean()
    abs_rel = np.mean(np.abs(y - x) / y)
    rmse = np.sqrt(((y - x) ** 2).mean())
    log_10 = np.abs(np.log10(y) - np.log10(x)).mean()
    return abs_rel, rmse, log_10, a1, a2, a3


def nyu_prediction_chain(pred_y, crop, border_crop_size=16):
    """pred_y [B,1,240,320]-sized network output already divided by 100 (utils.py:211) -> cropped [B,h,w] prediction:
    shrink to (240-b, 320-b), replicate-pad b/2, upsample x2 (all bilinear align_corners=True), clamp [0.4, 10], Eigen crop."""
    import torch
    import torch.nn.functional as F
    t = torch.from_numpy(np.ascontiguousarray(pred_y)).float()
    t = F.interpolate(t, (240 - border_crop_size, 320 - border_crop_size), mode="bilinear", align_corners=True)
    t = torch.nn.ReplicationPad2d(border_crop_size // 2)(t)
    t = F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    t = torch.clamp(t, min=0.4, max=10)
    return t[:, 0, crop[0]:crop[1] + 1, crop[2]:crop[3] + 1].numpy()
