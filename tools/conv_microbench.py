"""Time single decoder layers through the C ABI (development aid; also the target of rocprofv3 --pmc).
usage: python tools/conv_microbench.py [layer ...] [--iters N] [--batch B]
Layers are the config-2 (KITTI R18 640x192) dense convolutions, named by their state_dict index."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib, ops

# name: (C1, up1, C2, Cout, H, W, k, act)
LAYERS = {
    "0": (512, 1, 0, 256, 6, 20, 3, "elu"), "1": (256, 2, 256, 256, 12, 40, 3, "elu"),
    "5": (256, 1, 0, 128, 12, 40, 3, "elu"), "6": (128, 2, 128, 128, 24, 80, 3, "elu"),
    "9": (128, 1, 0, 64, 24, 80, 3, "elu"), "10": (64, 2, 64, 64, 48, 160, 3, "elu"),
    "13": (64, 1, 0, 32, 48, 160, 3, "elu"), "14": (32, 2, 64, 32, 96, 320, 3, "elu"),
    "14k2": (64, 2, 128, 32, 96, 320, 3, "elu"), "14k4": (128, 2, 256, 32, 96, 320, 3, "elu"),   # L14 with 2x / 4x the reduction: per-block fixed cost
    "h4": (256, 1, 0, 256, 12, 40, 1, "leaky"), "h3": (128, 1, 0, 128, 24, 80, 1, "leaky"),
    "h2": (64, 1, 0, 64, 48, 160, 1, "leaky"), "h1": (32, 1, 0, 32, 96, 320, 1, "leaky"),
}

ap = argparse.ArgumentParser()
ap.add_argument("layers", nargs="*", default=list(LAYERS))
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--act", default=None, help="override the activation (none | elu | leaky)")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name in args.layers:
    C1, up, C2, Cout, H, W, k, act = LAYERS[name]
    act = args.act or act
    x1 = torch.randn(args.batch, C1, H // up, W // up, device=dev)
    x2 = torch.randn(args.batch, C2, H, W, device=dev) if C2 else None
    w = torch.randn(Cout, C1 + C2, k, k, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    wp = ops.pack_weights(w)
    ww = ops.pack_weights_wino(w)
    for _ in range(3):
        ops._conv_fwd_raw(x1, x2, wp, b, Cout, k, "reflect", act, 0.1, up, ww)
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(args.iters):
        ops._conv_fwd_raw(x1, x2, wp, b, Cout, k, "reflect", act, 0.1, up, ww)
    recs = _lib.profile_end()
    fl = 2.0 * (C1 + C2) * k * k * Cout * args.batch * H * W
    tot = sum(r["ms"] for r in recs) / args.iters
    print("layer %-3s %4d+%-3d->%-3d %3dx%-3d k%d : %8.1f us  %6.1f TFLOP/s  [%s]" % (
        name, C1, C2, Cout, H, W, k, tot * 1e3, fl / tot / 1e9,
        ", ".join("%s %.1fus" % (r["kernel"], r["ms"] / r["calls"] * 1e3) for r in recs)))
