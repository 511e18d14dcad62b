"""CPU oracle (torch-CPU, differentiable) of the photometric loss stack — TEST INFRASTRUCTURE ONLY (see oracle/__init__).

Restates KITTI/layers.py:176-229 (BackprojectDepth, Project3D), :238-252 (get_smooth_loss), :281-311 (SSIM) and
KITTI/trainer.py:393-405 (compute_reprojection_loss), :352-372 (the grid_sample call).  Pinned by outputs AND gradients
of the reference's own layers module (tests/golden/photo_reference.npz, tests/golden/make_golden_photo.py);
compute_reprojection_loss is a method of the un-importable trainer and is restated on top of the pinned SSIM.
"""
import torch
import torch.nn.functional as F


def ssim(x, y):
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(x, 3, 1)
    mu_y = F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def compute_reprojection_loss(pred, target, use_ssim=True):
    l1 = torch.abs(target - pred).mean(1, True)
    if not use_ssim:
        return l1
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def backproject(depth, inv_K):
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W)], 0).unsqueeze(0).repeat(B, 1, 1)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(B, 1, -1) * cam
    return torch.cat([cam, torch.ones(B, 1, H * W)], 1)


def project(points, K, T, H, W, eps=1e-7):
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(B, 2, H, W).permute(0, 2, 3, 1)
    pix = torch.stack([pix[..., 0] / (W - 1), pix[..., 1] / (H - 1)], -1)
    return (pix - 0.5) * 2


def warp_frame(color, depth, K, inv_K, T):
    H, W = depth.shape[-2:]
    grid = project(backproject(depth, inv_K), K, T, H, W)
    return F.grid_sample(color, grid, padding_mode="border", align_corners=False)


def get_smooth_loss(disp, img, gamma=2):
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gamma * gix)).mean() + (gdy * torch.exp(-gamma * giy)).mean()


# ---- the trainer's orchestration (KITTI/trainer.py:329-560), torch-CPU, on top of the pinned operators above ----------

def disp_to_depth(disp, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled = min_disp + (max_disp - min_disp) * disp
    return scaled, 1 / scaled


def generate_images_pred(inputs, outputs, opt):
    for scale in opt.loss_scales:
        disp = outputs[("disp", scale)]
        if opt.v1_multiscale:
            source_scale = scale
        else:
            disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
            source_scale = 0
        _, depth = disp_to_depth(disp, opt.min_depth, opt.max_depth)
        outputs[("depth", 0, scale)] = depth
        for frame_id in opt.frame_ids[1:]:
            T = inputs["stereo_T"] if frame_id == "s" else outputs[("cam_T_cam", 0, frame_id)]
            outputs[("color", frame_id, scale)] = warp_frame(inputs[("color", frame_id, source_scale)], depth,
                                                             inputs[("K", source_scale)], inputs[("inv_K", source_scale)], T)
    if opt.use_depth_hints and "s" in opt.frame_ids[1:]:
        outputs[("color_depth_hint", "s", 0)] = warp_frame(inputs[("color", "s", 0)], inputs["depth_hint"], inputs[("K", 0)],
                                                           inputs[("inv_K", 0)], inputs["stereo_T"])
    return outputs


def compute_losses(inputs, outputs, opt):
    """compute_losses_hints with the tie-breaking noise left out (the tests pass tie_break_noise=0.0 to the product)."""
    losses, total = {}, 0
    dh_reproj = None
    if opt.use_depth_hints:
        dh_reproj = compute_reprojection_loss(outputs[("color_depth_hint", "s", 0)], inputs[("color", 0, 0)], not opt.no_ssim)
        dh_reproj = dh_reproj + 1000 * (1 - inputs["depth_hint_mask"])
    for scale in opt.loss_scales:
        source_scale = scale if opt.v1_multiscale else 0
        disp, color, target = outputs[("disp", scale)], inputs[("color", 0, scale)], inputs[("color", 0, source_scale)]
        reproj = torch.cat([compute_reprojection_loss(outputs[("color", f, scale)], target, not opt.no_ssim) for f in opt.frame_ids[1:]], 1)
        ident = torch.cat([compute_reprojection_loss(inputs[("color", f, source_scale)], target, not opt.no_ssim) for f in opt.frame_ids[1:]], 1)
        if opt.avg_reprojection:
            ident_l, reproj_l = ident.mean(1, keepdim=True), reproj.mean(1, keepdim=True)
        else:
            ident_l, reproj_l = torch.min(ident, 1, keepdim=True)[0], torch.min(reproj, 1, keepdim=True)[0]
        parts = [reproj_l, ident_l] + ([dh_reproj] if dh_reproj is not None else [])
        idxs = torch.argmin(torch.cat(parts, 1), 1, keepdim=True)
        mask = (idxs != 1).float()
        r = (reproj_l * mask).sum() / (mask.sum() + 1e-7)
        losses["reproj_loss/{}".format(scale)] = r
        outputs["identity_selection/{}".format(scale)] = 1 - mask
        dhl = 0
        if opt.use_depth_hints:
            dmask = (idxs == 2).float()
            dh = torch.log(torch.abs(inputs["depth_hint"] - outputs[("depth", 0, scale)]) + 1) * inputs["depth_hint_mask"]
            dhl = (dh * dmask).sum() / (dmask.sum() + 1e-7)
            losses["depth_hint_loss/{}".format(scale)] = dhl
        norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
        loss = r + dhl + opt.disparity_smoothness * get_smooth_loss(norm_disp, color) / (2 ** scale)
        total = total + loss
        losses["loss/{}".format(scale)] = loss
    losses["loss"] = total / len(opt.loss_scales)
    return losses
