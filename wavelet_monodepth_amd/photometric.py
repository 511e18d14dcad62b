"""Photometric loss stack of the KITTI trainer on the GPU, differentiable, through
libwmd_hip.so (csrc/wmd_photo.hip).  Names follow the reference:

    SSIM()(x, y)                                  KITTI/layers.py:281-311
    compute_reprojection_loss(pred, target)       KITTI/trainer.py:393-405 (0.85 SSIM + 0.15 L1, one kernel)
    warp_frame(color, depth, K, inv_K, T)         BackprojectDepth + Project3D + F.grid_sample(padding_mode="border")
                                                  (KITTI/layers.py:176-229, KITTI/trainer.py:352-372) fused
    get_smooth_loss(disp, img, gamma=2)           KITTI/layers.py:238-252
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
