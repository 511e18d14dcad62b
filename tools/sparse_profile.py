"""Per-kernel time of the sparse KITTI decoder (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wavelet_monodepth_amd import _lib, synth
from wavelet_monodepth_amd.kitti import SparseDepthWaveProgressiveDecoder
dev = torch.device("cuda:0")
chans = [64, 64, 128, 256, 512]
sp = synth.fill_state_dict(SparseDepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(1, 192, 640, chans, seed=1)]
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.15
for _ in range(3):
    sp(feats, thr)
torch.cuda.synchronize()
_lib.profile_begin()
n = 5
for _ in range(n):
    sp(feats, thr)
recs = _lib.profile_end()
print("thresh %.2f: %.3f ms of library kernels" % (thr, sum(r["ms"] for r in recs) / n))
for r in sorted(recs, key=lambda r: -r["ms"]):
    print("  %-40s calls %3d  %8.1f us/step  (%.1f us/call)" % (r["kernel"], r["calls"] // n, r["ms"] / n * 1e3, r["ms"] / r["calls"] * 1e3))
