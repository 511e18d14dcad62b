"""Time wmd_conv_wgrad on the decoder's layer shapes for every tile configuration (development aid).
usage: python tools/wgrad_microbench.py [r18|r50] [--cfgs 1,2,...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib

which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "r18"
cfgs = None
for a in sys.argv:
    if a.startswith("--cfgs="):
        cfgs = [int(v) for v in a[7:].split(",")]
# name: (B, C1, up, C2, Cout, H, W, k)
if which == "r18":
    B = 12
    L = {"L14": (32, 2, 64, 32, 96, 320, 3), "L13": (64, 1, 0, 32, 48, 160, 3), "L10": (64, 2, 64, 64, 48, 160, 3),
         "L9": (128, 1, 0, 64, 24, 80, 3), "L6": (128, 2, 128, 128, 24, 80, 3), "L5": (256, 1, 0, 128, 12, 40, 3),
         "L1": (256, 2, 256, 256, 12, 40, 3), "L0": (512, 1, 0, 256, 6, 20, 3),
         "h1": (32, 1, 0, 3, 96, 320, 3), "h2": (64, 1, 0, 3, 48, 160, 3), "h3": (128, 1, 0, 3, 24, 80, 3), "h4": (256, 1, 0, 3, 12, 40, 3)}
else:
    B = 8
    L = {"L14": (32, 2, 64, 32, 160, 512, 3), "L13": (64, 1, 0, 32, 80, 256, 3), "L10": (64, 2, 256, 64, 80, 256, 3),
         "L9": (128, 1, 0, 64, 40, 128, 3), "L6": (128, 2, 512, 128, 40, 128, 3), "L5": (256, 1, 0, 128, 20, 64, 3),
         "L1": (256, 2, 1024, 256, 20, 64, 3), "L0": (2048, 1, 0, 256, 10, 32, 3)}
dev = torch.device("cuda:0")
l = _lib.lib()
torch.manual_seed(0)
for name, (C1, up, C2, Cout, H, W, k) in L.items():
    x1 = torch.randn(B, C1, H // up, W // up, device=dev)
    x2 = torch.randn(B, C2, H, W, device=dev) if C2 else None
    dz = torch.randn(B, Cout, H, W, device=dev)
    dw = torch.empty(Cout, C1 + C2, k, k, device=dev)
    db = torch.empty(Cout, device=dev)
    res = []
    ref = None
    db_ref = None
    runs = [("direct", 0)] + [("direct", c) for c in (cfgs or [])] + [("wino", c) for c in range(0, l.wmd_conv_wgrad_num_configs() + 1)]
    for kind, cfg in runs:
        os.environ.pop("WMD_WGRAD_CFG", None)
        os.environ.pop("WMD_WGRAD_WINO_CFG", None)
        os.environ["WMD_WGRAD_WINO"] = "1" if kind == "wino" else "0"
        if cfg:
            os.environ["WMD_WGRAD_CFG" if kind == "direct" else "WMD_WGRAD_WINO_CFG"] = str(cfg)
        a = _lib.ConvWgradArgs(B=B, H=H, W=W, C1=C1, up1=up, C2=C2, Cout=Cout, ksize=k, pad_mode=1, x1=x1.data_ptr(),
                               x2=None if x2 is None else x2.data_ptr(), dz=dz.data_ptr(), dw=dw.data_ptr(), dbias=db.data_ptr(),
                               workspace=None, workspace_floats=0, tune_cfg=0, tune_nsplit=0)
        n = l.wmd_conv_wgrad_workspace_floats(C.byref(a))
        ws = torch.empty(max(n, 1), device=dev)
        a.workspace, a.workspace_floats = ws.data_ptr(), n
        s = torch.cuda.current_stream().cuda_stream
        if l.wmd_conv_wgrad(C.byref(a), s) != 0:
            continue
        torch.cuda.synchronize()
        if ref is None:
            ref = dw.clone()
        err = float((dw - ref).abs().max() / ref.abs().max())
        _lib.profile_begin()
        for _ in range(5):
            l.wmd_conv_wgrad(C.byref(a), s)
        recs = _lib.profile_end()
        main = [r for r in recs if "reduce" not in r["kernel"]][0]
        red = sum(r["ms"] for r in recs if "reduce" in r["kernel"]) / 5
        db_ref = db.clone() if cfg == 0 and kind == "direct" else db_ref
        err = max(err, float((db - db_ref).abs().max() / db_ref.abs().max()))
        res.append((main["ms"] / 5, red, cfg, main["kernel"], err))
    fl = 2.0 * (C1 + C2) * k * k * Cout * B * H * W
    print("%-4s %4d+%-4d->%-3d %3dx%-3d" % (name, C1, C2, Cout, H, W))
    seen = set()
    for ms, red, cfg, kern, err in sorted(res):
        if (kern, round(ms, 4)) in seen:
            continue
        seen.add((kern, round(ms, 4)))
        print("      cfg %2d %-40s %8.1f us (+%5.1f reduce) %6.1f TFLOP/s  err %.1e" % (cfg, kern, ms * 1e3, red * 1e3, fl / ms / 1e9, err))
os.environ.pop("WMD_WGRAD_CFG", None)
