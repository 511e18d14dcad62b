"""Backward of the stacked wavelet heads per decoder level (config 2 shapes): library kernels with the 3x3 stage on its own
kernels (wmd_head3x3_bwd) and on the generic dgrad / wgrad kernels (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import _lib, ops, synth
dev = torch.device("cuda:0")
B = 12
t = lambda a: torch.from_numpy(a).to(dev)
for (C, H, W, ll) in ((32, 96, 320, False), (64, 48, 160, False), (128, 24, 80, False), (256, 12, 40, True)):
    x = t(synth.normal((B, C, H, W), "x", 1)).requires_grad_(True)
    mk = lambda tag, mid, out: [t(a).requires_grad_(True) for a in synth.conv_params(tag + "1", mid, C, 1, 1)] + \
                               [t(a).requires_grad_(True) for a in synth.conv_params(tag + "3", out, mid, 3, 1)]
    hp, hn = mk("p", C, 3), mk("n", C, 3)
    hl = mk("l", C // 4, 1) if ll else None
    gy = t(synth.normal((B, 3, H, W), "g", 1))
    for new in (True, False):
        ops._HEAD_BWD = new
        ops._HEAD_BWD_MIN_PIXELS = 0
        ops._HEAD_BWD1_MIN_PIXELS = 0
        for it in range(4):
            if it == 3:
                _lib.profile_begin()
            yh, yl = ops.stacked_heads(x, hp, hn, 2.0, head_ll=hl, scale_ll=8.0, x_gate=("elu", 0.0))
            loss = (yh * gy).sum() + (yl.sum() if ll else 0.0)
            loss.backward()
        recs = _lib.profile_end()
        tot = sum(r["ms"] for r in recs)
        print("C %3d %3dx%3d own=%d: %.1f us total | %s" % (C, H, W, new, tot * 1e3, "  ".join("%s %.1f" % (r["kernel"].replace("_kernel", ""), r["ms"] * 1e3) for r in sorted(recs, key=lambda r: -r["ms"])[:9])))
