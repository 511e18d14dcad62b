"""Photometric loss stack of the KITTI trainer on the GPU, differentiable, through
libwmd_hip.so (csrc/wmd_photo.hip).  Names follow the reference:

    SSIM()(x, y)                                  KITTI/layers.py:281-311
    compute_reprojection_loss(pred, target)       KITTI/trainer.py:393-405 (0.85 SSIM + 0.15 L1, one kernel)
    warp_frame(color, depth, K, inv_K, T)         BackprojectDepth + Project3D + F.grid_sample(padding_mode="border")
                                                  (KITTI/layers.py:176-229, KITTI/trainer.py:352-372) fused
    get_smooth_loss(disp, img, gamma=2)           KITTI/layers.py:238-252
    generate_images_pred / compute_loss_masks / compute_losses   the trainer's orchestration (KITTI/trainer.py:329-560)
No CPU fallback: CPU tensors raise.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from ._lib import check, current_stream, ptr


def _gpu(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.WmdError("the photometric operators run on the GPU only (got a %s tensor)" % t.device)
        if t.dtype != torch.float32:
            raise _lib.WmdError("float32 tensors expected (got %s)" % t.dtype)


class _SsimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mode, w_ssim, w_l1):
        x, y = x.contiguous(), y.contiguous()
        B, Cc, H, W = x.shape
        out = torch.empty((B, Cc if mode == 0 else 1, H, W), device=x.device, dtype=torch.float32)
        check(_lib.lib().wmd_ssim_fwd(ptr(x), ptr(y), ptr(out), B, Cc, H, W, mode, w_ssim, w_l1, current_stream()), "wmd_ssim_fwd")
        ctx.save_for_backward(x, y)
        ctx.cfg = (mode, w_ssim, w_l1)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        mode, w_ssim, w_l1 = ctx.cfg
        B, Cc, H, W = x.shape
        l = _lib.lib()
        need_x, need_y = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = torch.empty_like(x) if need_x else None
        dy = torch.empty_like(y) if need_y else None
        if need_x or need_y:
            n = l.wmd_ssim_bwd_workspace_floats(B, Cc, H, W)
            ws = torch.empty(n, device=x.device, dtype=torch.float32)
            check(l.wmd_ssim_bwd(ptr(x), ptr(y), ptr(g.contiguous()), ptr(dx), ptr(dy), B, Cc, H, W, mode, w_ssim, w_l1, ptr(ws), n,
                                 current_stream()), "wmd_ssim_bwd")
        return dx, dy, None, None, None


class SSIM(nn.Module):
    """Layer to compute the SSIM loss between a pair of images: clamp((1 - SSIM(x, y)) / 2, 0, 1), [B,C,H,W]."""

    def forward(self, x, y):
        _gpu(x, y)
        return _SsimFn.apply(x, y, 0, 1.0, 0.0)


def compute_reprojection_loss(pred, target, use_ssim=True, no_ssim=False):
    """[B,1,H,W]: mean_c |target - pred| when SSIM is off, else 0.85 * mean_c SSIM-loss + 0.15 * mean_c L1."""
    _gpu(pred, target)
    if no_ssim or not use_ssim:
        return _SsimFn.apply(pred, target, 1, 0.0, 1.0)
    return _SsimFn.apply(pred, target, 1, 0.85, 0.15)


class _WarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, depth, K, inv_K, T, eps):
        color, depth, K, inv_K, T = (t.contiguous() for t in (color, depth, K, inv_K, T))
        B, Cc, Hs, Ws = color.shape
        H, W = depth.shape[-2:]
        out = torch.empty((B, Cc, H, W), device=color.device, dtype=torch.float32)
        a = _lib.WarpArgs(B=B, C=Cc, H=H, W=W, Hs=Hs, Ws=Ws, eps=float(eps), src=ptr(color), depth=ptr(depth), K=ptr(K),
                          inv_K=ptr(inv_K), T=ptr(T))
        check(_lib.lib().wmd_warp_fwd(C.byref(a), ptr(out), current_stream()), "wmd_warp_fwd")
        ctx.save_for_backward(color, depth, K, inv_K, T)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        color, depth, K, inv_K, T = ctx.saved_tensors
        B, Cc, Hs, Ws = color.shape
        H, W = depth.shape[-2:]
        l = _lib.lib()
        a = _lib.WarpArgs(B=B, C=Cc, H=H, W=W, Hs=Hs, Ws=Ws, eps=ctx.eps, src=ptr(color), depth=ptr(depth), K=ptr(K),
                          inv_K=ptr(inv_K), T=ptr(T))
        ddepth = torch.empty_like(depth)
        dT = torch.empty_like(T)
        n = l.wmd_warp_bwd_workspace_floats(C.byref(a))
        ws = torch.empty(n, device=color.device, dtype=torch.float32)
        check(l.wmd_warp_bwd(C.byref(a), ptr(g.contiguous()), ptr(ddepth), ptr(dT), ptr(ws), n, current_stream()), "wmd_warp_bwd")
        return None, ddepth, None, None, dT, None


def warp_frame(color, depth, K, inv_K, T, eps=1e-7):
    """color [B,C,Hs,Ws] source frame, depth [B,1,H,W], K / inv_K / T [B,4,4] -> the source frame resampled into the target
    view [B,C,H,W].  Differentiable w.r.t. depth and T (the colour frame and the intrinsics are constants of the loss)."""
    _gpu(color, depth, K, inv_K, T)
    if color.requires_grad or K.requires_grad or inv_K.requires_grad:
        raise _lib.WmdError("warp_frame differentiates w.r.t. depth and T only")
    return _WarpFn.apply(color, depth, K, inv_K, T, eps)


class _SmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img, gamma):
        disp, img = disp.contiguous(), img.contiguous()
        B, _, H, W = disp.shape
        l = _lib.lib()
        n = l.wmd_smooth_workspace_floats(B, H, W)
        ws = torch.empty(n, device=disp.device, dtype=torch.float32)
        out = torch.empty(1, device=disp.device, dtype=torch.float32)
        check(l.wmd_smooth_fwd(ptr(disp), ptr(img), ptr(out), B, img.shape[1], H, W, float(gamma), ptr(ws), n, current_stream()),
              "wmd_smooth_fwd")
        ctx.save_for_backward(disp, img)
        ctx.gamma = float(gamma)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        disp, img = ctx.saved_tensors
        B, _, H, W = disp.shape
        dd = torch.empty_like(disp)
        check(_lib.lib().wmd_smooth_bwd(ptr(disp), ptr(img), ptr(g.reshape(1).contiguous()), ptr(dd), B, img.shape[1], H, W, ctx.gamma,
                                        current_stream()), "wmd_smooth_bwd")
        return dd, None, None


def get_smooth_loss(disp, img, gamma=2):
    """Edge-aware smoothness of a (mean-normalised) disparity map; scalar tensor, differentiable w.r.t. disp."""
    _gpu(disp, img)
    if disp.shape[1] != 1:
        raise _lib.WmdError("disp must be [B,1,H,W]")
    return _SmoothFn.apply(disp, img, gamma)


# ---------------------------------------------------------------------------------------------
# The trainer's loss orchestration on top of the operators above (KITTI/trainer.py:329-560).  Plain tensor bookkeeping:
# which frames are warped, the per-pixel minimum over frames, auto-masking against the identity reprojection, the
# depth-hint selection, the per-scale weighting.  Option names follow the reference's `opt`.
# ---------------------------------------------------------------------------------------------

class LossOptions:
    """The fields of the reference's option object that the loss uses (KITTI/options.py defaults)."""

    def __init__(self, height=192, width=640, frame_ids=(0, -1, 1), loss_scales=(0, 1, 2, 3), min_depth=0.1, max_depth=100.0,
                 v1_multiscale=False, disable_automasking=False, avg_reprojection=False, no_ssim=False, use_depth_hints=False,
                 disparity_smoothness=1e-3, scales=(0, 1, 2, 3)):
        self.height, self.width = height, width
        self.frame_ids, self.loss_scales = list(frame_ids), list(loss_scales)
        self.scales = list(scales)      # --scales: the normaliser of the total loss (trainer.py:47,557), separate from --loss_scales
        self.min_depth, self.max_depth = min_depth, max_depth
        self.v1_multiscale, self.disable_automasking, self.avg_reprojection = v1_multiscale, disable_automasking, avg_reprojection
        self.no_ssim, self.use_depth_hints, self.disparity_smoothness = no_ssim, use_depth_hints, disparity_smoothness


def generate_images_pred(inputs, outputs, opt):
    """trainer.py:329-392 (pose_model_type != "posecnn"): every ("disp", s) -> full-resolution depth -> the source frames
    warped into the target view, stored as outputs[("depth", 0, s)] and outputs[("color", frame_id, s)]."""
    from . import ops
    for scale in opt.loss_scales:
        disp = outputs[("disp", scale)]
        if opt.v1_multiscale:
            source_scale = scale
            min_disp, max_disp = 1 / opt.max_depth, 1 / opt.min_depth
            depth = 1 / (min_disp + (max_disp - min_disp) * disp)
        else:
            source_scale = 0
            _, depth = ops.upsample_bilinear(disp, (opt.height, opt.width), align_corners=False,
                                             depth_range=(opt.min_depth, opt.max_depth))   # upsample + disp_to_depth, one kernel
        outputs[("depth", 0, scale)] = depth
        for frame_id in opt.frame_ids[1:]:
            T = inputs["stereo_T"] if frame_id == "s" else outputs[("cam_T_cam", 0, frame_id)]
            outputs[("color", frame_id, scale)] = warp_frame(inputs[("color", frame_id, source_scale)], depth,
                                                             inputs[("K", source_scale)], inputs[("inv_K", source_scale)], T)
            if not opt.disable_automasking:
                outputs[("color_identity", frame_id, scale)] = inputs[("color", frame_id, source_scale)]
    if opt.use_depth_hints and "s" in opt.frame_ids[1:]:
        outputs[("color_depth_hint", "s", 0)] = warp_frame(inputs[("color", "s", 0)], inputs["depth_hint"], inputs[("K", 0)],
                                                           inputs[("inv_K", 0)], inputs["stereo_T"])
    return outputs


def compute_loss_masks(reprojection_loss, identity_reprojection_loss, depth_hint_reprojection_loss):
    """trainer.py:425-458."""
    depth_hint_loss_mask = None
    if identity_reprojection_loss is None:
        reprojection_loss_mask = torch.ones_like(reprojection_loss)
        if depth_hint_reprojection_loss is not None:
            idxs = torch.argmin(torch.cat([reprojection_loss, depth_hint_reprojection_loss], dim=1), dim=1, keepdim=True)
            depth_hint_loss_mask = (idxs == 1).float()
    else:
        parts = [reprojection_loss, identity_reprojection_loss]
        if depth_hint_reprojection_loss is not None:
            parts.append(depth_hint_reprojection_loss)
        idxs = torch.argmin(torch.cat(parts, dim=1), dim=1, keepdim=True)
        reprojection_loss_mask = (idxs != 1).float()   # the auto-mask has index 1
        if depth_hint_reprojection_loss is not None:
            depth_hint_loss_mask = (idxs == 2).float()
    return reprojection_loss_mask, depth_hint_loss_mask


def compute_losses(inputs, outputs, opt, tie_break_noise=None):
    """trainer.py:460-560 (compute_losses_hints; without depth hints it is Monodepth2's loss with the minimum taken as we
    go).  `tie_break_noise`: None draws the reference's randn * 1e-5 on the identity loss; a float (0.0 in the tests)
    multiplies a fixed zero field instead."""
    losses = {}
    total_loss = 0
    depth_hint_reproj_loss = None
    if opt.use_depth_hints:
        depth_hint_reproj_loss = compute_reprojection_loss(outputs[("color_depth_hint", "s", 0)], inputs[("color", 0, 0)],
                                                           no_ssim=opt.no_ssim)
        depth_hint_reproj_loss = depth_hint_reproj_loss + 1000 * (1 - inputs["depth_hint_mask"])
    for scale in opt.loss_scales:
        source_scale = scale if opt.v1_multiscale else 0
        disp = outputs[("disp", scale)]
        color = inputs[("color", 0, scale)]
        target = inputs[("color", 0, source_scale)]
        reproj = torch.cat([compute_reprojection_loss(outputs[("color", f, scale)], target, no_ssim=opt.no_ssim)
                            for f in opt.frame_ids[1:]], 1)
        if opt.disable_automasking:
            raise NotImplementedError   # as the reference (trainer.py:506)
        ident = torch.cat([compute_reprojection_loss(inputs[("color", f, source_scale)], target, no_ssim=opt.no_ssim)
                           for f in opt.frame_ids[1:]], 1)
        if opt.avg_reprojection:
            identity_loss = ident.mean(1, keepdim=True)
            reprojection_loss = reproj.mean(1, keepdim=True)
        else:
            identity_loss, _ = torch.min(ident, dim=1, keepdim=True)
            reprojection_loss, _ = torch.min(reproj, dim=1, keepdim=True)
        if tie_break_noise is None:   # "add random numbers to break ties" (trainer.py:517-520)
            identity_loss = identity_loss + torch.randn(identity_loss.shape, device=identity_loss.device) * 0.00001
        reprojection_loss_mask, depth_hint_loss_mask = compute_loss_masks(reprojection_loss, identity_loss, depth_hint_reproj_loss)
        reprojection_loss = (reprojection_loss * reprojection_loss_mask).sum() / (reprojection_loss_mask.sum() + 1e-7)
        outputs["identity_selection/{}".format(scale)] = (1 - reprojection_loss_mask).float()
        losses["reproj_loss/{}".format(scale)] = reprojection_loss
        depth_hint_loss = 0
        if opt.use_depth_hints:
            dh = torch.log(torch.abs(inputs["depth_hint"] - outputs[("depth", 0, scale)]) + 1) * inputs["depth_hint_mask"]
            depth_hint_loss = (dh * depth_hint_loss_mask).sum() / (depth_hint_loss_mask.sum() + 1e-7)
            outputs["depth_hint_pixels/{}".format(scale)] = depth_hint_loss_mask
            losses["depth_hint_loss/{}".format(scale)] = depth_hint_loss
        loss = reprojection_loss + depth_hint_loss
        norm_disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
        loss = loss + opt.disparity_smoothness * get_smooth_loss(norm_disp, color) / (2 ** scale)
        total_loss = total_loss + loss
        losses["loss/{}".format(scale)] = loss
    losses["loss"] = total_loss / len(getattr(opt, "scales", opt.loss_scales))   # self.num_scales = len(opt.scales), trainer.py:47,557
    return losses
