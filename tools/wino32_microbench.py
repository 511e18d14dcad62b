"""conv_wino32_kernel against the 16x16x4 kernels on the trunk layers of BASELINE config 2 (development aid).

For every layer: the best of the existing configurations as chosen by the on-device tuner restricted to them, then every
conv_wino32 configuration x split-K, each checked against the direct kernel's output and timed (hipEvents, min / median).
usage: python tools/wino32_microbench.py [layer ...] [--iters N] [--batch B] [--size r18|r50]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib, ops, tuner

# name: (C1, up1, C2, Cout, H, W)
R18 = {"0": (512, 1, 0, 256, 6, 20), "1": (256, 2, 256, 256, 12, 40), "5": (256, 1, 0, 128, 12, 40),
       "6": (128, 2, 128, 128, 24, 80), "9": (128, 1, 0, 64, 24, 80), "10": (64, 2, 64, 64, 48, 160),
       "13": (64, 1, 0, 32, 48, 160), "14": (32, 2, 64, 32, 96, 320)}
R50 = {"1": (256, 2, 1024, 256, 20, 64), "6": (128, 2, 512, 128, 40, 128), "10": (64, 2, 256, 64, 80, 256),
       "14": (32, 2, 64, 32, 160, 512), "13": (64, 1, 0, 32, 80, 256), "9": (128, 1, 0, 64, 40, 128)}

ap = argparse.ArgumentParser()
ap.add_argument("layers", nargs="*", default=["14", "10", "13", "6", "9", "1", "5"])
ap.add_argument("--iters", type=int, default=15)
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--size", default="r18")
ap.add_argument("--ksplits", default="1,2,4")
ap.add_argument("--cfgs", default="", help="semicolon-separated template argument lists of the wino32 configurations to run (default: all)")
ap.add_argument("--no-old", action="store_true", help="skip the sweep over the existing configurations")
ap.add_argument("--mask-which", default="both", help="both | in | out")
ap.add_argument("--mask", type=float, default=0.0, help="block-sparse mode: density of the i.i.d. coarse seed mask whose dilations are the in / out masks")
args = ap.parse_args()
LAYERS = R18 if args.size == "r18" else R50
dev = torch.device("cuda:0")
torch.manual_seed(0)
l = _lib.lib()
names = tuner.config_names()
stream = lambda: torch.cuda.current_stream().cuda_stream
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def run(a, ws_cache={}):
    n = l.wmd_conv_fwd_workspace_floats(C.byref(a))
    if n not in ws_cache:
        ws_cache[n] = torch.empty(max(n, 1), device=dev)
    a.workspace, a.workspace_floats = ws_cache[n].data_ptr(), n
    return l.wmd_conv_fwd(C.byref(a), stream())


def timeit(a, iters):
    ts = []
    for _ in range(iters):
        e0.record()
        st = run(a)
        e1.record()
        e1.synchronize()
        if st != 0:
            return None
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


for name in args.layers:
    C1, up, C2, Cout, H, W = LAYERS[name]
    B = args.batch
    x1 = torch.randn(B, C1, H // up, W // up, device=dev)
    x2 = torch.randn(B, C2, H, W, device=dev) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    wp, ww = ops.pack_weights(w), ops.pack_weights_wino(w)
    y = torch.empty(B, Cout, H, W, device=dev)

    im = om = None
    if args.mask > 0:
        import torch.nn.functional as F
        seed = (torch.rand(B, 1, H // 2, W // 2, device=dev) < args.mask).float()
        im = F.interpolate(F.max_pool2d(seed, 3, 1, 1), scale_factor=2).to(torch.uint8).reshape(B, H, W).contiguous()   # 2x2-constant
        om = F.max_pool2d(F.interpolate(seed, scale_factor=2), 3, 1, 1).to(torch.uint8).reshape(B, H, W).contiguous()
        print("   masks: in %.2f out %.2f" % (float(im.float().mean()), float(om.float().mean())))

    def mk(cfg, ks):
        if im is not None:
            return _lib.ConvArgs(B=B, H=H, W=W, C1=C1, up1=up, C2=C2, Cout=Cout, ksize=3, pad_mode=ops.PAD["reflect"], act=ops.ACT["elu"],
                                 slope=0.0, x1=x1.data_ptr(), x2=None if x2 is None else x2.data_ptr(), wp=wp.data_ptr(),
                                 bias=b.data_ptr(), y=y.data_ptr(), workspace=None, workspace_floats=0, tune_cfg=cfg,
                                 tune_ksplit=ks, wp_wino=ww.data_ptr(), in_mask=im.data_ptr() if args.mask_which != "out" else None,
                                 out_mask=om.data_ptr() if args.mask_which != "in" else None, in_mask_2x2=1)
        return _lib.ConvArgs(B=B, H=H, W=W, C1=C1, up1=up, C2=C2, Cout=Cout, ksize=3, pad_mode=ops.PAD["reflect"], act=ops.ACT["elu"],
                             slope=0.0, x1=x1.data_ptr(), x2=None if x2 is None else x2.data_ptr(), wp=wp.data_ptr(),
                             bias=b.data_ptr(), y=y.data_ptr(), workspace=None, workspace_floats=0, tune_cfg=cfg,
                             tune_ksplit=ks, wp_wino=ww.data_ptr())

    fl = 2.0 * (C1 + C2) * 9 * Cout * B * H * W
    # reference output: a direct (non-Winograd) configuration chosen by the library's model
    direct = [i + 1 for i, n in enumerate(names) if (n.startswith("conv_wino_kernel") if im is not None else n.endswith(",9>"))]
    ref = None
    for cfg in direct:
        y.zero_()
        a = mk(cfg, 1)
        if run(a) == 0:
            torch.cuda.synchronize()
            ref = y.clone()
            break
    print("== layer %s: %d(up%d)+%d -> %d @ %dx%d, batch %d, %.2f GFLOP" % (name, C1, up, C2, Cout, H, W, B, fl / 1e9))
    # best existing configuration (everything but the wino32 family), a quick sweep
    best = None
    for i, n in enumerate(names):
        if args.no_old:
            break
        if n.startswith("conv_wino32") or not (n.endswith(",9>") or n.startswith("conv_wino_kernel")):
            continue
        for ks in (1, 2, 4, 8):
            a = mk(i + 1, ks)
            if run(a) != 0:
                continue
            r = timeit(a, 3)
            if r and (best is None or r[0] < best[0]):
                best = (r[0], n, ks, i + 1)
    if best is not None:
      r = timeit(mk(best[3], best[2]), args.iters)
      print("   old best  %-40s ks%d : min %7.1f med %7.1f us  %6.1f TFLOP/s(alg)" % (best[1], best[2], r[0], r[1], fl / r[1] / 1e6))
    for i, n in enumerate(names):
        if not n.startswith("conv_wino32"):
            continue
        if args.cfgs and n.split("<")[1].rstrip(">") not in args.cfgs.split(";"):
            continue
        for ks in [int(k) for k in args.ksplits.split(",")]:
            y.fill_(0.0 if im is not None else float("nan"))
            a = mk(i + 1, ks)
            st = run(a)
            if st != 0:
                continue
            torch.cuda.synchronize()
            err = float((y - ref).abs().max() / ref.abs().max())
            r = timeit(a, args.iters)
            print("   %-50s ks%d : min %7.1f med %7.1f us  %6.1f TFLOP/s(alg)  max rel err %.2e %s" % (
                n, ks, r[0], r[1], fl / r[1] / 1e6, err, "" if err < 5e-5 else "  <-- MISMATCH"))
