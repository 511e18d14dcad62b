"""Host cost per forward of the graph-replayed decoders at batch 1 (is the replay loop host-bound?)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wavelet_monodepth_amd import synth
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder
dev = torch.device("cuda:0")
chans = [64, 64, 128, 256, 512]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sp = synth.fill_state_dict(SparseDepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
dn = DepthWaveProgressiveDecoder(np.array(chans)).to(dev)
dn.load_state_dict(sp.state_dict())
feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(B, 192, 640, chans, seed=1)]
sp.enable_graph(True); dn.enable_graph(True)
def run(fn, n=300):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
with torch.no_grad():
    print("dense  B=%d: host %.1f us / forward, loop %.1f us" % ((B,) + run(lambda: dn(feats))))
    print("sparse B=%d: host %.1f us / forward, loop %.1f us" % ((B,) + run(lambda: sp(feats, 0.15))))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): sp(feats, 0.15)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
