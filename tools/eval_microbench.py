"""Time the evaluation chain on a KITTI-sized batch (development aid): 16 images, 192x640 predictions, 375x1242 ground truth."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib, evaluation as ev

dev = torch.device("cuda:0")
B = 16
disp = torch.rand(B, 192, 640, device=dev) * 0.3 + 0.01
gt = torch.rand(B, 375, 1242, device=dev) * 90
gt[torch.rand_like(gt) < 0.7] = 0
for _ in range(3):
    ev.kitti_metrics(disp, gt)
torch.cuda.synchronize()
_lib.profile_begin()
for _ in range(10):
    ev.kitti_metrics(disp, gt)
recs = _lib.profile_end()
tot = sum(r["ms"] for r in recs) / 10
print("kitti_metrics batch %d: %.3f ms (%.1f us / image): %s" % (B, tot, tot * 1e3 / B, ", ".join(
    "%s %.1f us %.0f GB/s" % (r["kernel"], r["ms"] / r["calls"] * 1e3, r["bytes"] / r["ms"] / 1e6) for r in recs)))
l, r = torch.rand(B, 192, 640, device=dev), torch.rand(B, 192, 640, device=dev)
for _ in range(3):
    ev.flip_postprocess(l, r)
torch.cuda.synchronize()
_lib.profile_begin()
for _ in range(10):
    ev.flip_postprocess(l, r)
recs = _lib.profile_end()
print(", ".join("%s %.1f us %.0f GB/s" % (r["kernel"], r["ms"] / r["calls"] * 1e3, r["bytes"] / r["ms"] / 1e6) for r in recs))
