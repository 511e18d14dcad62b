"""CPU: the photometric oracle (oracle/photo_ref.py) against outputs and gradients of the reference's own KITTI/layers.py
(tests/golden/photo_reference.npz, made by tests/golden/make_golden_photo.py)."""
import numpy as np
import torch

from oracle import photo_ref as P
from wavelet_monodepth_amd import synth
from util import load_golden, photo_case


def t(a, g=False):
    return torch.from_numpy(a.copy()).requires_grad_(g)


def test_ssim_and_gradients():
    g = load_golden("photo_reference.npz")
    tgt, src, *_ = photo_case()
    x, y = t(src, True), t(tgt, True)
    s = P.ssim(x, y)
    w = torch.from_numpy(synth.uniform(tuple(s.shape), "ph_w", 31, 0.0, 1.0).astype(np.float32))
    (s * w).sum().backward()
    np.testing.assert_allclose(s.detach().numpy(), g["ssim"], atol=2e-6)
    np.testing.assert_allclose(x.grad.numpy(), g["ssim_dx"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y.grad.numpy(), g["ssim_dy"], rtol=1e-4, atol=2e-5)


def test_warp_and_gradients():
    g = load_golden("photo_reference.npz")
    tgt, src, depth, K, inv_K, T = photo_case()
    d, Tt = t(depth, True), t(T, True)
    pix = P.project(P.backproject(d, t(inv_K)), t(K), Tt, *depth.shape[-2:])
    np.testing.assert_allclose(pix.detach().numpy(), g["pix_coords"], atol=2e-6)
    out = P.warp_frame(t(src), d, t(K), t(inv_K), Tt)
    w = torch.from_numpy(synth.uniform(tuple(out.shape), "ph_w", 31, 0.0, 1.0).astype(np.float32))
    (out * w).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), g["warp"], atol=2e-5)
    np.testing.assert_allclose(d.grad.numpy(), g["warp_ddepth"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(Tt.grad.numpy(), g["warp_dT"], rtol=1e-3, atol=1e-3)


def test_smooth_loss_and_gradient():
    g = load_golden("photo_reference.npz")
    tgt, src, depth, *_ = photo_case()
    disp = t((1.0 / depth).astype(np.float32), True)
    sm = P.get_smooth_loss(disp, t(tgt))
    sm.backward()
    np.testing.assert_allclose(float(sm), float(g["smooth"]), rtol=1e-6)
    np.testing.assert_allclose(disp.grad.numpy(), g["smooth_ddisp"], rtol=1e-5, atol=1e-8)
