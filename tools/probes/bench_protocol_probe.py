"""What the driver's short protocol (--steps 20 --warmup 5) costs against steady state (development aid): the replayed
config-2 forward timed over K steps after W warm-ups, with and without a per-step event record in the timed loop."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wavelet_monodepth_amd import synth, tuner
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
tuner.preload(os.path.join(ROOT, "profiles", "r05_tune_cache.json"))
dev = torch.device("cuda:0")
R18 = [64, 64, 128, 256, 512]
dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev).eval()
feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(12, 192, 640, R18, seed=1)]
dec.enable_graph(True)


def run(K, W, marks, idle_s=0.0):
    with torch.no_grad():
        if idle_s:
            torch.cuda.synchronize()
            time.sleep(idle_s)           # let the clocks fall back, like a fresh process after its set-up
        for _ in range(W):
            dec(feats)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] if marks else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if marks:
            ev[0].record()
        for k in range(K):
            dec(feats)
            if marks:
                ev[k + 1].record()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / K * 1e3
    per = [ev[k].elapsed_time(ev[k + 1]) for k in range(K)] if marks else []
    return el, per


with torch.no_grad():
    dec(feats)
for K, W in ((20, 5), (50, 10), (300, 5)):
    for marks in (True, False):
        for idle in (0.0, 0.5):
            el, per = run(K, W, marks, idle)
            extra = "  first 5 steps: " + " ".join("%.3f" % v for v in per[:5]) + "  last: %.3f" % per[-1] if per else ""
            print("K=%3d W=%2d marks=%d idle %.1fs : %.4f ms/step%s" % (K, W, marks, idle, el, extra), flush=True)
