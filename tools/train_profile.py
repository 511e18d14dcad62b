"""Per-kernel time of one decoder training step (fwd + bwd) on synthetic config-2/3 shapes (development aid)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from wavelet_monodepth_amd import _lib, synth
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--chans", default="64,64,128,256,512")
ap.add_argument("--height", type=int, default=192)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--update", action="store_true", help="touch the weights after every step like an optimizer would (the "
                "weight images are then rebuilt each step) and print the wall-clock time per step too")
ap.add_argument("--nyu", action="store_true", help="NYUv2 DecoderWave at DenseNet161 widths, 640x480 (BASELINE config 5)")
args = ap.parse_args()
chans = [int(c) for c in args.chans.split(",")]
dev = torch.device("cuda:0")
if args.nyu:
    from wavelet_monodepth_amd.nyu import DecoderWave
    chans, args.height, args.width = [96, 96, 192, 384, 2208], 480, 640
    args.batch = 4 if args.batch == 12 else args.batch
    dec = synth.fill_state_dict(DecoderWave(enc_features=chans), seed=9).to(dev)
    feats = [torch.from_numpy(synth.normal((args.batch, c, args.height >> (k + 1), args.width >> (k + 1)), "nyu%d" % k, 9)).to(dev).requires_grad_(True)
             for k, c in enumerate(chans)]
else:
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
    feats = [torch.from_numpy(f).to(dev).requires_grad_(True) for f in synth.encoder_features(args.batch, args.height, args.width, chans, seed=1)]


params = list(dec.parameters())


def step():
    out = dec(feats)
    loss = sum(out[("disp", s)].mean() for s in range(4))
    loss.backward()
    if args.update:
        with torch.no_grad():
            torch._foreach_mul_(params, 1.0)     # bumps the version counters: every memoised weight image is stale


for _ in range(3):
    step()
torch.cuda.synchronize()
if args.update:
    import time
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
    print("decoder fwd+bwd + weight touch: %.3f ms wall-clock per step" % wall)
_lib.profile_begin()
n = 5
for _ in range(n):
    step()
recs = _lib.profile_end()
tot = sum(r["ms"] for r in recs) / n
print("decoder fwd+bwd: %.3f ms of library kernels per step (batch %d, %dx%d)" % (tot, args.batch, args.width, args.height))
for r in sorted(recs, key=lambda r: -r["ms"]):
    print("  %-44s calls/step %3d  %8.3f ms/step  %6.1f TFLOP/s" % (r["kernel"], r["calls"] // n, r["ms"] / n,
                                                                   r["flops"] / max(r["ms"], 1e-9) / 1e9))
