"""Diagnose gradient mismatches vs the oracle at full config shapes: error histogram + location of the worst elements."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from oracle import decoder_ref as R
from wavelet_monodepth_amd import synth
from util import R50, kitti_feats

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "r50half"
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
if which == "r50half":
    chans, B, H, W, seed = R50, 2, 160, 512, 4
elif which == "r50":
    chans, B, H, W, seed = R50, 1, 320, 1024, 3
else:
    chans, B, H, W, seed = [64, 64, 128, 256, 512], 2, 192, 640, 5
dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(chans)), seed=seed).to(dev)
feats = kitti_feats(B, H, W, chans, seed=seed)
sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in dec.state_dict().items()}
fc = [f.clone().requires_grad_(True) for f in feats]
loss_kind = sys.argv[2] if len(sys.argv) > 2 else "sq"


def loss_fn(out):
    if loss_kind == "sq":
        return sum((out[("disp", s)] ** 2).mean() for s in range(4))
    if loss_kind == "coef":   # no clamp in the path: the raw coefficient planes
        return sum((out[("wavelets", s, b)] ** 2).mean() for s in range(4) for b in ("LH", "HL", "HH")) + (out[("wavelets", 3, "LL")] ** 2).mean()
    return sum(out[("disp", s)].mean() for s in range(4))


ref = R.kitti_wave_decoder(fc, sd)
loss_fn(ref).backward()
fg = [f.to(dev).requires_grad_(True) for f in feats]
og = dec(fg)
loss_fn(og).backward()
for s in range(4):
    d = ref[("disp", s)].detach()
    print("disp%d: clamped-at-0 %.4f clamped-at-1 %.4f" % (s, float((d <= 0).float().mean()), float((d >= 1).float().mean())))


def report(name, a, b):
    a, b = a.detach().cpu().double(), b.detach().double()
    scale = float(b.abs().max())
    err = (a - b).abs() / max(scale, 1e-30)
    bad = (err > 1e-4)
    msg = "%-34s max %.2e  n(>1e-4) %d / %d  n(>1e-3) %d" % (name, float(err.max()), int(bad.sum()), err.numel(), int((err > 1e-3).sum()))
    if bad.any():
        idx = torch.nonzero(err == err.max())[0].tolist()
        msg += "  worst at %s gpu %.4e ref %.4e" % (idx, float(a[tuple(idx)]), float(b[tuple(idx)]))
    print(msg)


for k, (a, b) in enumerate(zip(fg, fc)):
    report("dfeat%d" % k, a.grad, b.grad)
for n, p in dec.named_parameters():
    report(n, p.grad, sd[n].grad)
