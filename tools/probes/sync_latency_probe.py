"""How much of a short timed region is the host waking up from torch.cuda.synchronize()?  K replays of the headline forward timed
(a) with a blocking synchronize at the end, (b) spinning on an event first (development probe for bench.py's protocol)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from wavelet_monodepth_amd import tuner
tuner.preload(os.path.join(ROOT, "profiles", bench.TUNE_CACHE))
dev = torch.device("cuda:0")
dec, feats = bench.build_model(dev)
with torch.no_grad():
    for _ in range(10):
        dec(feats)
    torch.cuda.synchronize()
    for K in (20, 50, 200):
        res = {}
        for mode in ("sync", "spin", "sync", "spin"):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                dec(feats)
            if mode == "spin":
                ev = torch.cuda.Event()
                ev.record()
                while not ev.query():
                    pass
            torch.cuda.synchronize()
            res.setdefault(mode, []).append((time.perf_counter() - t0) / K * 1e3)
        print("K=%d: blocking synchronize %s ms/step, event spin + synchronize %s" % (K, ["%.4f" % v for v in res["sync"]], ["%.4f" % v for v in res["spin"]]))
