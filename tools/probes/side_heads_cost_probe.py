"""How much of the replayed config-2 step do the side-stream heads (levels 4-2) and the level-1 head cost?  Replays the
captured segments with some of them left out (outputs are then stale: timing only).  Development aid."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wavelet_monodepth_amd import synth, tuner
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
tuner.preload(os.path.join(ROOT, "profiles", "r03_tune_cache.json"))
dev = torch.device("cuda:0")
R18 = [64, 64, 128, 256, 512]
dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev).eval()
feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(12, 192, 640, R18, seed=1)]
dec.enable_graph(True)


class Skip:
    def replay(self):
        pass


def run(n=300):
    with torch.no_grad():
        for _ in range(5):
            dec(feats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            dec(feats)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    dec(feats)
ent = list(next(iter(dec._segments.values())))
heads = ent[1]
full = dict(heads)
for rep in range(2):
    for name, skip in (("everything", ()), ("without H4-H2 (side stream)", (4, 3, 2)), ("without H1", (1,)), ("trunk only", (4, 3, 2, 1))):
        for i in full:
            heads[i] = Skip() if i in skip else full[i]
        print("%-30s %.4f ms/step" % (name, run()), flush=True)
for i in full:
    heads[i] = full[i]
