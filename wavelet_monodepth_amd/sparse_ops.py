"""Operator layer of the threshold-gated sparse path (batch 1, inference only; the reference forbids
training with it, KITTI/trainer.py:35-36).  Thin wrappers over the wmd_mask_* / wmd_sparse_conv entry points;
see include/wmd.h for the layout decision (dense zero-initialised activations + masks + compacted pixel lists).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ACT, PAD, check, current_stream, ptr


def _on_gpu(*tensors):
    """No CPU implementation in the package: every tensor handed to the library must live on the GPU."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.WmdError("wavelet_monodepth_amd sparse ops run on the MI355X only (got a %s tensor)" % t.device)


def minmax(x):
    """-> device tensor [min, max] of x (depth_decoder.py:308)."""
    _on_gpu(x)
    out = torch.empty(2, device=x.device, dtype=torch.float32)
    x = x.contiguous()
    check(_lib.lib().wmd_minmax(ptr(x), x.numel(), ptr(out), current_stream()), "wmd_minmax")
    return out


def mask_threshold(yh, mm, thresh_ratio):
    """yh [1,1,3,h,w] (or [3,h,w]) -> uint8 mask [h,w]: max_b |yh_b| > (max-min)*ratio (:308-309)."""
    _on_gpu(yh, mm)
    h, w = yh.shape[-2:]
    yh = yh.contiguous()
    mask = torch.empty((h, w), device=yh.device, dtype=torch.uint8)
    check(_lib.lib().wmd_mask_threshold(ptr(yh), ptr(mm), float(thresh_ratio), ptr(mask), h, w, current_stream()),
          "wmd_mask_threshold")
    return mask


def dilate_multi(mask, specs):
    """mask uint8 [h,w]; specs = [(up, radius), ...] -> list of uint8 masks [h*up, w*up] (one launch)."""
    _on_gpu(mask)
    h, w = mask.shape
    outs = [torch.empty((h * up, w * up), device=mask.device, dtype=torch.uint8) for up, _ in specs]
    arr = (_lib.DilateSpec * len(specs))(*[_lib.DilateSpec(up, r, ptr(o)) for (up, r), o in zip(specs, outs)])
    check(_lib.lib().wmd_mask_dilate_multi(ptr(mask), h, w, arr, len(specs), current_stream()), "wmd_mask_dilate_multi")
    return outs


def mask_level(yl, yh, thresh_ratio, specs):
    """minmax(yl) -> threshold(yh) -> every dilated variant, one launch (depth_decoder.py:308-319).
    specs = [(up, radius), ...]; (1, 0) is the thresholded mask itself.  Bit-identical to the three separate calls."""
    _on_gpu(yl, yh)
    h, w = yh.shape[-2:]
    yl, yh = yl.contiguous(), yh.contiguous()
    outs = [torch.empty((h * up, w * up), device=yh.device, dtype=torch.uint8) for up, _ in specs]
    arr = (_lib.DilateSpec * len(specs))(*[_lib.DilateSpec(up, r, ptr(o)) for (up, r), o in zip(specs, outs)])
    check(_lib.lib().wmd_mask_level(ptr(yl), yl.numel(), ptr(yh), float(thresh_ratio), h, w, arr, len(specs),
                                    current_stream()), "wmd_mask_level")
    return outs


def compact_multi(masks):
    """uint8 masks -> (list of int32 coordinate lists [npix capacity], int32 tensor of counts [n]) in one launch;
    raster order, counts stay on the device."""
    _on_gpu(*masks)
    n = len(masks)
    nnz = torch.empty(n, device=masks[0].device, dtype=torch.int32)
    coords = [torch.empty(m.numel(), device=m.device, dtype=torch.int32) for m in masks]
    arr = (_lib.CompactSpec * n)(*[_lib.CompactSpec(ptr(m), m.numel(), ptr(c), nnz.data_ptr() + 4 * i)
                                   for i, (m, c) in enumerate(zip(masks, coords))])
    check(_lib.lib().wmd_mask_compact_multi(arr, n, current_stream()), "wmd_mask_compact_multi")
    return coords, nnz


def sparse_conv(y, x1, wp, bias, cout, ksize, out_coords, out_nnz, max_out, x2=None, up1=1, in_mask=None,
                pad="reflect", act="none", slope=0.0, out_scale=1.0, c1=None, c1_off=0, wp2=None, bias2=None,
                c1_off2=0, split_waves=0):
    """Gather-GEMM convolution on the active pixels; writes y [Cout,H,W] in place at those pixels."""
    _on_gpu(y, x1, x2, wp, bias, in_mask, out_coords, wp2, bias2)
    Cout_, H, W = y.shape
    assert Cout_ == cout
    c1tot = x1.shape[0]
    c1 = c1tot if c1 is None else c1
    a = _lib.SparseConvArgs(H=H, W=W, C1=c1, up1=up1, C1tot=c1tot, c1_off=c1_off, C2=0 if x2 is None else x2.shape[0],
                            Cout=cout, ksize=ksize, pad_mode=PAD[pad], act=ACT[act], slope=float(slope),
                            x1=ptr(x1), x2=ptr(x2), in_mask=ptr(in_mask), out_coords=ptr(out_coords),
                            out_nnz=out_nnz, max_out=int(max_out), wp=ptr(wp), bias=ptr(bias), wp2=ptr(wp2),
                            bias2=ptr(bias2), c1_off2=c1_off2, out_scale=float(out_scale), y=ptr(y),
                            split_waves=int(split_waves))
    check(_lib.lib().wmd_sparse_conv(C.byref(a), current_stream()), "wmd_sparse_conv")
    return y
