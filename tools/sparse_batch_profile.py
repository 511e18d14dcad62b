"""Per-kernel time of one batched sparse decode (development aid). usage: python tools/sparse_batch_profile.py [density] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from wavelet_monodepth_amd import _lib, synth
from wavelet_monodepth_amd.kitti import SparseDepthWaveProgressiveDecoder

p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.10
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
chans = [64, 64, 128, 256, 512]
sp = synth.fill_state_dict(SparseDepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
batch = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(K, 192, 640, chans, seed=10)]
fb = {i: (torch.from_numpy(synth.uniform((K, h, w), "bdens%d" % i, 3, 0.0, 1.0)) < p).to(torch.uint8).to(dev)
      for i, (h, w) in zip((3, 2, 1), ((12, 40), (24, 80), (48, 160)))}
thr = 0.05
if p < 0:            # negative "density" = free-running threshold -p
    thr, fb = -p, None
for _ in range(3):
    sp(batch, thr, _force_masks=fb)
torch.cuda.synchronize()
_lib.profile_begin()
n = 5
for _ in range(n):
    sp(batch, thr, _force_masks=fb)
recs = _lib.profile_end()
print("batched sparse decode, %d frames, injected density %.2f: %.3f ms of library kernels" % (K, p, sum(r["ms"] for r in recs) / n))
for r in sorted(recs, key=lambda r: -r["ms"]):
    print("  %-44s calls %3d  %8.3f ms" % (r["kernel"], r["calls"] // n, r["ms"] / n))
