"""Heads on a second stream beside the trunk, EAGER launches (a replayed hipGraph serialises a captured fork): wall time per
forward of BASELINE config 2 for graph replay, eager one stream, eager two streams at several side-stream priorities, each with
the main stream at default and at high priority.  usage: python tools/probes/overlap_probe.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
from wavelet_monodepth_amd import tuner
tuner.preload(os.path.join(ROOT, "profiles", bench.TUNE_CACHE))
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")


def timed(dec, feats, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with torch.no_grad(), ctx:
        for _ in range(10):
            out = dec(feats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = dec(feats)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
    return dt, out


os.environ["WMD_BENCH_GRAPH"] = "1"
dec, feats = bench.build_model(dev)
ms, ref = timed(dec, feats)
ref = {k: v.clone() for k, v in ref.items()}
print("graph replay              : %.4f ms" % ms)
dec.enable_graph(False)
ms, _ = timed(dec, feats)
print("eager, one stream         : %.4f ms" % ms)
for prio in (0, 1, 2, -1):
    for main_hi in (False, True):
        os.environ["WMD_OVERLAP_PRIO"] = str(prio)
        dec.overlap_heads = 2
        dec._side_stream = None
        try:
            st = torch.cuda.Stream(priority=-1) if main_hi else None
            ms, out = timed(dec, feats, st)
        except Exception as e:
            print("side prio %d main_hi %s: %r" % (prio, main_hi, e))
            continue
        err = max(float((out[k] - ref[k]).abs().max()) for k in ref)
        print("eager, heads on side stream (prio %2d), main %s : %.4f ms   max abs diff vs replay %.2e" % (prio, "high" if main_hi else "dflt", ms, err))
