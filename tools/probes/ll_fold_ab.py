"""In-process A/B of the coarsest level's low-pass head as a fused third launch vs its own operators (development aid):
hipGraph replay time of the config-2 decoder forward, then the eager per-kernel breakdown of both forms."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from wavelet_monodepth_amd import ops, synth, tuner, _lib
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
tuner.preload(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "profiles", "r03_tune_cache.json"))
dev = torch.device("cuda:0")
R18 = [64, 64, 128, 256, 512]
dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev).eval()
feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(12, 192, 640, R18, seed=1)]
dec.enable_graph(True)
def run(n):
    with torch.no_grad():
        for _ in range(5): dec(feats)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): dec(feats)
        t_enq = (time.perf_counter() - t0) / n * 1e3       # host time to enqueue: == the total when the host is the limit
        torch.cuda.synchronize()
    run.enqueue_ms = t_enq
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for fold in (False, True):
        ops._LL_FOLD = fold
        dec.enable_graph(True)      # drops the captured segments
        print("LL_FOLD=%d  %.4f ms/step (host enqueue %.4f ms/step)" % (fold, run(300), run.enqueue_ms), flush=True)
dec.enable_graph(False)
for fold in (False, True):
    ops._LL_FOLD = fold
    with torch.no_grad():
        for _ in range(3): dec(feats)
        torch.cuda.synchronize()
        _lib.profile_begin()
        for _ in range(10): dec(feats)
        recs = _lib.profile_end()
    print("LL_FOLD=%d eager library kernels: %.1f us/step" % (fold, sum(r["ms"] for r in recs) / 10 * 1e3))
    for r in sorted(recs, key=lambda r: -r["ms"]):
        if "head" in r["kernel"] or "fused" in r["kernel"] or r["ms"] / 10 < 0.02:
            print("    %-46s %2d x %7.1f us" % (r["kernel"][:46], r["calls"] // 10, r["ms"] / r["calls"] * 1e3))
