"""Per-call time of the 12 sparse_conv launches (and the mask kernels) of one sparse KITTI decode, each replayed REPS
times inside a hipGraph so that host launch overhead is out of the picture (development aid).
usage: python tools/sparse_microbench.py [thresh=0.15] [--nyu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wavelet_monodepth_amd import synth, sparse_ops as S
from wavelet_monodepth_amd.kitti import SparseDepthWaveProgressiveDecoder

dev = torch.device("cuda:0")
chans = [64, 64, 128, 256, 512]
sp = synth.fill_state_dict(SparseDepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(1, 192, 640, chans, seed=1)]
thr = float([a for a in sys.argv[1:] if not a.startswith("--")][0]) if len([a for a in sys.argv[1:] if not a.startswith("--")]) else 0.15
REPS = 20

calls = []
keep = []
names = ["sparse_conv", "mask_level", "compact_multi"]
orig = {n: getattr(S, n) for n in names}


def rec(n):
    def f(*a, **k):
        r = orig[n](*a, **k)
        calls.append((n, a, k))
        keep.append(r)      # raw pointers (the pixel counts) must stay valid
        return r
    return f


for n in names:
    setattr(S, n, rec(n))
for _ in range(2):
    calls.clear()
    out = sp(feats, thr)
keep.append(out)
for n in names:
    setattr(S, n, orig[n])
torch.cuda.synchronize()
print("thresh %.2f densities %s" % (thr, " ".join("%.2f" % float(out[("wavelet_mask", s)].float().mean()) for s in (2, 1, 0))))


def time_call(n, a, k):
    fn = lambda: orig[n](*a, **k)
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(REPS):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REPS) * 1e3


tot = 0.0
for n, a, k in calls:
    us = time_call(n, a, k)
    tot += us
    if n == "sparse_conv":
        y, x1 = a[0], a[1]
        desc = "Cin %3d%s -> Cout %3d k%d  %3dx%-3d" % (k.get("c1") or x1.shape[0], "+%d" % k["x2"].shape[0] if k.get("x2") is not None else "",
                                                       a[4], a[5], y.shape[1], y.shape[2])
    else:
        desc = ""
    print("  %-14s %-40s %7.1f us" % (n, desc, us))
print("sum %.1f us" % tot)
