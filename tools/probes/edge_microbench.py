"""Encoder edge: upconv(4,0) on the encoder's last pre-activation with the ReLU applied on load, against
ReLU (a PyTorch elementwise kernel, what the encoder would run) + the ordinary convolution (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import ops, synth
dev = torch.device("cuda:0")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, n=30):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, (B, C, H, W, Cout) in {"R18 640x192 b12": (12, 512, 6, 20, 256), "R50 1024x320 b8": (8, 2048, 10, 32, 256)}.items():
    pre = torch.randn(B, C, H, W, device=dev)
    w, b = [torch.from_numpy(a).to(dev) for a in synth.conv_params("w", Cout, C, 3, 1)]
    with torch.no_grad():
        act = torch.relu(pre)
        ta = t(lambda: ops.conv2d_fused(act, w, b, pad="reflect", act="elu"))
        tr = t(lambda: torch.relu(pre))
        tp = t(lambda: ops.conv2d_pre_activated(pre, (None, None, "leaky", 0.0), w, b, pad="reflect", act="elu"))
        err = float((ops.conv2d_pre_activated(pre, (None, None, "leaky", 0.0), w, b, pad="reflect", act="elu") -
                     ops.conv2d_fused(act, w, b, pad="reflect", act="elu")).abs().max())
    print("%s: relu %.1f us + tuned conv %.1f us = %.1f us | ReLU-on-load conv %.1f us | max abs diff %.1e" % (name, tr, ta, tr + ta, tp, err))
