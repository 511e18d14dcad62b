"""Time the wavelet heads of config 2 level by level (development aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib, ops

dev = torch.device("cuda:0")
B = 12
for C, H, W in [(256, 12, 40), (128, 24, 80), (64, 48, 160), (32, 96, 320)]:
    mid = torch.randn(B, 2 * C, H, W, device=dev)
    wp = torch.randn(3, C, 3, 3, device=dev) * 0.05
    wn = torch.randn(3, C, 3, 3, device=dev) * 0.05
    b = torch.randn(3, device=dev)
    for _ in range(3):
        ops.head3x3_nograd(mid, C, 0, wp, b, C, wn, b, mode=2, scale=2.0)
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(20):
        ops.head3x3_nograd(mid, C, 0, wp, b, C, wn, b, mode=2, scale=2.0)
    recs = _lib.profile_end()
    us = recs[0]["ms"] / recs[0]["calls"] * 1e3
    fl = 2 * 2 * 27 * C * B * H * W
    print("head3x3 C=%3d %3dx%-3d: %7.1f us  %5.1f TFLOP/s" % (C, H, W, us, fl / us / 1e6))

# full level (both heads + IDWT), one-launch vs two-launch form (WMD_TWO_LAUNCH_HEAD=1)
for C, H, W in [(256, 12, 40), (128, 24, 80), (64, 48, 160), (32, 96, 320)]:
    x = torch.randn(B, C, H, W, device=dev)
    yl = torch.randn(B, 1, H, W, device=dev)
    mk = lambda: (torch.randn(C, C, 1, 1, device=dev) * 0.1, torch.randn(C, device=dev), torch.randn(3, C, 3, 3, device=dev) * 0.05,
                  torch.randn(3, device=dev))
    hp, hn = mk(), mk()
    for _ in range(3):
        ops.head_fused_level_nograd(x, hp, hn, 2.0, yl, 0.5, True)
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(20):
        ops.head_fused_level_nograd(x, hp, hn, 2.0, yl, 0.5, True)
    recs = _lib.profile_end()
    print("level C=%3d %3dx%-3d: %s" % (C, H, W, ", ".join("%s %.1f us" % (r["kernel"], r["ms"] / r["calls"] * 1e3) for r in recs)))

# the unfused inference form (stacked 1x1 conv through the tuned trunk kernel + head3x3 + IDWT), per level
import numpy as np
from wavelet_monodepth_amd import synth
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
from wavelet_monodepth_amd.wavelets import IDWT
dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array([64, 64, 128, 256, 512])), seed=1).to(dev)
idwt = IDWT().to(dev)
with torch.no_grad():
    for i, (C, H, W) in zip((4, 3, 2, 1), [(256, 12, 40), (128, 24, 80), (64, 48, 160), (32, 96, 320)]):
        x = torch.randn(B, C, H, W, device=dev)
        yl = torch.randn(B, 1, H, W, device=dev)
        def run():
            _, yh = dec.get_coefficients(x, scale=i, return_ll=False)
            return idwt((yl, [yh]))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        _lib.profile_begin()
        for _ in range(20):
            run()
        recs = _lib.profile_end()
        tot = sum(r["ms"] for r in recs) / 20 * 1e3
        print("unfused level %d C=%3d: %.1f us total: %s" % (i, C, tot, ", ".join("%s %.1f" % (r["kernel"], r["ms"] / 20 * 1e3) for r in recs)))
