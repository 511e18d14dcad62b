"""Golden vectors for the photometric loss stack from the REFERENCE's own KITTI/layers.py (importable: torch + numpy only):
SSIM, BackprojectDepth + Project3D (+ F.grid_sample as the trainer calls it, KITTI/trainer.py:352-372) and
get_smooth_loss — outputs and gradients on seeded inputs.  Run in the build container only:

    python tests/golden/make_golden_photo.py      # writes tests/golden/photo_reference.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/KITTI")
from wavelet_monodepth_amd import synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import photo_case  # noqa: E402  (inputs shared with the tests)
import layers as RL  # noqa: E402  (the reference's KITTI/layers.py)


def main():
    tgt, src, depth, K, inv_K, T = photo_case()
    B, _, H, W = tgt.shape
    t = lambda a, g=False: torch.from_numpy(a.copy()).requires_grad_(g)
    out = {}
    # SSIM + gradients w.r.t. both images
    x, y = t(src, True), t(tgt, True)
    s = RL.SSIM()(x, y)
    wgt = torch.from_numpy(synth.uniform(tuple(s.shape), "ph_w", 31, 0.0, 1.0).astype(np.float32))
    (s * wgt).sum().backward()
    out["ssim"], out["ssim_dx"], out["ssim_dy"] = s.detach().numpy(), x.grad.numpy(), y.grad.numpy()
    # warp + gradients w.r.t. depth and T
    d, Tt = t(depth, True), t(T, True)
    cam = RL.BackprojectDepth(B, H, W)(d, t(inv_K))
    pix = RL.Project3D(B, H, W)(cam, t(K), Tt)
    warped = F.grid_sample(t(src), pix, padding_mode="border")
    (warped * wgt).sum().backward()
    out["warp"], out["warp_ddepth"], out["warp_dT"] = warped.detach().numpy(), d.grad.numpy(), Tt.grad.numpy()
    out["pix_coords"] = pix.detach().numpy()
    # smoothness + gradient
    disp = t((1.0 / depth).astype(np.float32), True)
    sm = RL.get_smooth_loss(disp, t(tgt))
    sm.backward()
    out["smooth"], out["smooth_ddisp"] = sm.detach().numpy(), disp.grad.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "photo_reference.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).mean()))


if __name__ == "__main__":
    main()
