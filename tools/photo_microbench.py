"""Time the photometric loss operators at the KITTI training size (batch 12, 192x640): forward + backward, library kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from wavelet_monodepth_amd import _lib, photometric as ph

dev = torch.device("cuda:0")
B, H, W = 12, 192, 640
pred = torch.rand(B, 3, H, W, device=dev, requires_grad=True)
tgt = torch.rand(B, 3, H, W, device=dev)
src = torch.rand(B, 3, H, W, device=dev)
depth = (torch.rand(B, 1, H, W, device=dev) * 30 + 2).requires_grad_(True)
K = torch.tensor([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], device=dev).repeat(B, 1, 1)
inv_K = torch.linalg.inv(K).contiguous()
T = torch.eye(4, device=dev).repeat(B, 1, 1)
T[:, 0, 3] = 0.3
T.requires_grad_(True)
disp = torch.rand(B, 1, H, W, device=dev, requires_grad=True)


def step():
    loss = ph.compute_reprojection_loss(ph.warp_frame(src, depth, K, inv_K, T), tgt).mean() + ph.compute_reprojection_loss(pred, tgt).mean() \
        + 1e-3 * ph.get_smooth_loss(disp, tgt)
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
_lib.profile_begin()
for _ in range(10):
    step()
recs = _lib.profile_end()
tot = sum(r["ms"] for r in recs) / 10
print("photometric step (warp + 2x reprojection + smoothness, fwd+bwd), batch %d %dx%d: %.3f ms of library kernels" % (B, H, W, tot))
for r in sorted(recs, key=lambda r: -r["ms"]):
    print("  %-26s calls/step %2d  %7.1f us/call  %6.0f GB/s (algorithmic bytes)" % (
        r["kernel"], r["calls"] // 10, r["ms"] / r["calls"] * 1e3, r["bytes"] / r["ms"] / 1e6))
