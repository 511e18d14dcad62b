"""Evaluation arithmetic on the GPU: what KITTI/evaluate_depth.py and NYUv2/utils.py do per
image in numpy on the host, for whole batches resident in HBM, through libwmd_hip.so (csrc/wmd_eval.hip).

    kitti_metrics(pred_disp, gt_depth, ...)      evaluate_depth.py:268-307 + compute_errors (:50-68)
    flip_postprocess(l_disp, r_disp_raw)         evaluate_depth.py:71-79 with the [:, :, ::-1] of :204 fused
    compute_errors(pred, gt)                     evaluate_depth.py:50-68 on prepared arrays
    compute_errors_nyu(pred, gt)                 NYUv2/utils.py:85-98
    nyu_prediction(pred_y, crop)                 NYUv2/utils.py:213-226,247-250
No CPU fallback: CPU tensors raise.
"""
import ctypes as C

import torch

from . import _lib, ops
from ._lib import check, current_stream, ptr

KITTI_NAMES = ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")
NYU_NAMES = ("abs_rel", "rmse", "log_10", "a1", "a2", "a3")
MIN_DEPTH, MAX_DEPTH, STEREO_SCALE_FACTOR = 1e-3, 80.0, 5.4


def _gpu(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.WmdError("evaluation runs on the GPU only (got a %s tensor)" % t.device)
        if t.dtype != torch.float32:
            raise _lib.WmdError("evaluation expects float32 tensors (got %s)" % t.dtype)


def kitti_metrics(pred_disp, gt_depth, eval_split="eigen", eval_stereo=False, disable_median_scaling=False,
                  pred_depth_scale_factor=1.0, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH):
    """pred_disp [B,h,w] (network resolution), gt_depth [B,H,W] -> (errors [B,7] in KITTI_NAMES order, ratios [B] or
    None, n_valid [B]).  Options follow evaluate_depth.py:263-270: stereo evaluation disables median scaling and scales
    by 5.4."""
    _gpu(pred_disp, gt_depth)
    if eval_stereo:
        disable_median_scaling, pred_depth_scale_factor = True, STEREO_SCALE_FACTOR
    l = _lib.lib()
    pred_disp, gt_depth = pred_disp.contiguous(), gt_depth.contiguous()
    B, h, w = pred_disp.shape
    Bg, H, W = gt_depth.shape
    if B != Bg:
        raise _lib.WmdError("batch mismatch: %d predictions, %d ground-truth maps" % (B, Bg))
    n = l.wmd_eval_workspace_floats(B, H, W)
    ws = torch.empty(n, device=pred_disp.device, dtype=torch.float32)
    out = torch.empty((B, 9), device=pred_disp.device, dtype=torch.float32)
    a = _lib.EvalKittiArgs(B=B, h=h, w=w, H=H, W=W, min_depth=float(min_depth), max_depth=float(max_depth),
                           mask_mode=1 if eval_split == "eigen" else 0, pred_scale=float(pred_depth_scale_factor),
                           median_scaling=0 if disable_median_scaling else 1, pred_disp=ptr(pred_disp), gt_depth=ptr(gt_depth),
                           out=ptr(out), workspace=ptr(ws), workspace_floats=n)
    check(l.wmd_eval_kitti(C.byref(a), current_stream()), "wmd_eval_kitti")
    ratios = None
    if not disable_median_scaling:
        med = ws[2 * B * H * W + B:2 * B * H * W + 3 * B].view(B, 2)
        ratios = med[:, 1] / med[:, 0]
    return out[:, :7], ratios, out[:, 8].to(torch.int64)


def _errors9(pred, gt):
    _gpu(pred, gt)
    if pred.shape != gt.shape:
        raise _lib.WmdError("shape mismatch %s vs %s" % (tuple(pred.shape), tuple(gt.shape)))
    pred, gt = pred.contiguous(), gt.contiguous()
    B = pred.shape[0] if pred.dim() > 1 else 1
    out = torch.empty((B, 9), device=pred.device, dtype=torch.float32)
    check(_lib.lib().wmd_eval_errors(ptr(pred), ptr(gt), B, pred.numel() // B, ptr(out), current_stream()), "wmd_eval_errors")
    return out


def compute_errors(gt, pred):
    """KITTI compute_errors per leading-dimension item: [B,7] (argument order of the reference: gt first).  The reference
    calls it once per image and averages the rows afterwards (evaluate_depth.py:305-313), which is what the rows are for."""
    return _errors9(pred, gt)[:, :7]


def compute_errors_nyu(pred, gt, per_image=False):
    """NYUv2/utils.py:85-98: (abs_rel, rmse, log_10, a1, a2, a3) as a [6] tensor — ONE reduction over every element of the
    arrays handed in, as the reference computes them over its whole concatenated test set (rmse = sqrt of the GLOBAL mean
    squared error: averaging per-image rmse values afterwards gives a different, non-comparable number).
    per_image=True: one row per leading-dimension item, [B,6], for per-frame diagnostics."""
    if per_image:
        o = _errors9(pred, gt)
        return torch.stack([o[:, 0], o[:, 2], o[:, 7], o[:, 4], o[:, 5], o[:, 6]], 1)
    o = _errors9(pred.reshape(1, -1), gt.reshape(1, -1))[0]
    return torch.stack([o[0], o[2], o[7], o[4], o[5], o[6]])


def flip_postprocess(l_disp, r_disp_raw):
    """Monodepth-v1 post-processing; r_disp_raw is the prediction for the mirrored image, not flipped back."""
    _gpu(l_disp, r_disp_raw)
    l_disp, r_disp_raw = l_disp.contiguous(), r_disp_raw.contiguous()
    B, h, w = l_disp.shape
    out = torch.empty_like(l_disp)
    check(_lib.lib().wmd_flip_postprocess(ptr(l_disp), ptr(r_disp_raw), ptr(out), B, h, w, current_stream()), "wmd_flip_postprocess")
    return out


def nyu_prediction(pred_y, crop, border_crop_size=16):
    """NYUv2/utils.py:213-226,247-250 on the GPU: pred_y [B,1,H,W] (already / 100) -> [B, crop rows, crop cols]."""
    _gpu(pred_y)
    t = ops.upsample_bilinear(pred_y, (240 - border_crop_size, 320 - border_crop_size), align_corners=True)
    t = torch.nn.functional.pad(t, (border_crop_size // 2,) * 4, mode="replicate")
    t = ops.upsample_bilinear(t, (2 * t.shape[2], 2 * t.shape[3]), align_corners=True)
    t = torch.clamp(t, min=0.4, max=10)
    return t[:, 0, crop[0]:crop[1] + 1, crop[2]:crop[3] + 1]
