"""Cycle stamps inside conv_wino32_kernel blocks (development aid; needs the -DWMD_STAMPS build of the library:
tools/probes/build_variant.sh stamps "-DWMD_STAMPS" wmd_conv_wino32 wmd_conv_fwd wmd_conv_wino32q
-> tools/probes/_build/libwmd_stamps.so, loaded through WMD_LIB_PATH).

For each (layer, configuration, ksplit): one launch with a debug buffer, then per phase the median / p10 / p90 cycles over
the blocks, separately for the first wave of blocks (those that start within half a block time of the earliest) and the rest.
usage: WMD_LIB_PATH=tools/probes/_build/libwmd_stamps.so python tools/probes/stamps_probe.py [layer:cfg:ks ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib, ops, tuner

R18 = {"0": (512, 1, 0, 256, 6, 20), "1": (256, 2, 256, 256, 12, 40), "5": (256, 1, 0, 128, 12, 40),
       "6": (128, 2, 128, 128, 24, 80), "9": (128, 1, 0, 64, 24, 80), "10": (64, 2, 64, 64, 48, 160),
       "13": (64, 1, 0, 32, 48, 160), "14": (32, 2, 64, 32, 96, 320)}
PHASES = ["args", "tables", "offsets", "descr", "issue0", "wait0", "main", "otrans", "xch", "store"]
jobs = sys.argv[1:] or ["14:8,32,2,8:1", "10:8,32,2,8:1", "9:8,32,2,8:1", "1:6,40,2,8:4", "6:8,16,1,8:1", "13:8,32,2,8:1"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
l = _lib.lib()
names = tuner.config_names()
B = int(os.environ.get("STAMPS_BATCH", "12"))
for job in jobs:
    name, cfgs, ks = job.split(":")
    C1, up, C2, Cout, H, W = R18[name]
    cfg = names.index("conv_wino32_kernel<%s>" % cfgs) + 1
    x1 = torch.randn(B, C1, H // up, W // up, device=dev)
    x2 = torch.randn(B, C2, H, W, device=dev) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    wp, ww = ops.pack_weights(w), ops.pack_weights_wino(w)
    y = torch.empty(B, Cout, H, W, device=dev)
    a = _lib.ConvArgs(B=B, H=H, W=W, C1=C1, up1=up, C2=C2, Cout=Cout, ksize=3, pad_mode=ops.PAD["reflect"], act=ops.ACT["elu"],
                      slope=0.0, x1=x1.data_ptr(), x2=None if x2 is None else x2.data_ptr(), wp=wp.data_ptr(), bias=b.data_ptr(),
                      y=y.data_ptr(), workspace=None, workspace_floats=0, tune_cfg=cfg, tune_ksplit=int(ks), wp_wino=ww.data_ptr())
    n = l.wmd_conv_fwd_workspace_floats(C.byref(a))
    ws = torch.empty(max(n, 1), device=dev)
    a.workspace, a.workspace_floats = ws.data_ptr(), n
    dbg = torch.zeros(1 << 16, 12, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    os.environ.pop("WMD_DBG_PTR", None)
    for _ in range(3):
        assert l.wmd_conv_fwd(C.byref(a), st) == 0
    torch.cuda.synchronize()
    os.environ["WMD_DBG_PTR"] = hex(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert l.wmd_conv_fwd(C.byref(a), st) == 0
    e1.record()
    torch.cuda.synchronize()
    os.environ.pop("WMD_DBG_PTR", None)
    d = dbg.cpu()
    d = d[d[:, 0] != 0]
    t = d[:, :11].double()
    xcc = d[:, 11]
    start = torch.zeros(len(d), dtype=torch.float64)       # (the cycle counters of the XCDs are not synchronised: per-XCC time base)
    span = 0.0
    for i in range(8):
        m = xcc == i
        if int(m.sum()):
            start[m] = t[m][:, 0] - t[m][:, 0].min()
            span = max(span, float(t[m][:, 10].max() - t[m][:, 0].min()))
    dur = t[:, 10] - t[:, 0]
    print("== layer %s cfg <%s> ks%s batch %d [%s]: %d blocks stamped, launch %.1f us; block duration median %.0f cycles, longest XCC span %.0f cycles"
          % (name, cfgs, ks, B, os.environ.get("STAMPS_TAG", ""), len(d), e0.elapsed_time(e1) * 1e3, float(dur.median()), span))
    first = start < float(dur.median()) * 0.5
    for label, sel in (("first round", first), ("later rounds", ~first)):
        if int(sel.sum()) == 0:
            continue
        seg = t[sel][:, 1:11] - t[sel][:, 0:10]
        q = torch.quantile(seg, torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64), dim=0)
        print("   %-12s (%4d blocks; start offsets %.0f..%.0f)" % (label, int(sel.sum()), float(start[sel].min()), float(start[sel].max())))
        print("      " + " ".join("%8s" % p for p in PHASES) + "    total")
        for qi, qn in enumerate(("p10", "med", "p90")):
            print("  %s " % qn + " ".join("%8.0f" % float(v) for v in q[qi]) + " %8.0f" % float(q[qi].sum()))
    print("   blocks per XCC:", [int((xcc == i).sum()) for i in range(8)])
