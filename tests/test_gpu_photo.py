"""GPU: photometric loss stack (csrc/wmd_photo.hip through the C ABI) against the reference's own layers (goldens) and
the differentiable CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import photo_ref as P
from wavelet_monodepth_amd import _lib, photometric as ph, synth
from util import load_golden, photo_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def t(a, dev=None, g=False):
    x = torch.from_numpy(np.ascontiguousarray(a).copy())
    if dev is not None:
        x = x.to(dev)
    return x.requires_grad_(g)


def test_ssim_vs_reference_golden(dev):
    g = load_golden("photo_reference.npz")
    tgt, src, *_ = photo_case()
    x, y = t(src, dev, True), t(tgt, dev, True)
    s = ph.SSIM()(x, y)
    w = t(synth.uniform(tuple(s.shape), "ph_w", 31, 0.0, 1.0).astype(np.float32), dev)
    (s * w).sum().backward()
    # sigma = E[x^2] - mu^2 cancels on these smoothed frames: the summation order (fma contraction on the GPU) shows up at
    # 1e-5; the bar is north_star's 1e-4 relative
    np.testing.assert_allclose(s.detach().cpu().numpy(), g["ssim"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["ssim_dx"], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(y.grad.cpu().numpy(), g["ssim_dy"], rtol=2e-3, atol=2e-3)


def test_warp_vs_reference_golden(dev):
    g = load_golden("photo_reference.npz")
    tgt, src, depth, K, inv_K, T = photo_case()
    d, Tt = t(depth, dev, True), t(T, dev, True)
    out = ph.warp_frame(t(src, dev), d, t(K, dev), t(inv_K, dev), Tt)
    w = t(synth.uniform(tuple(out.shape), "ph_w", 31, 0.0, 1.0).astype(np.float32), dev)
    (out * w).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["warp"], atol=1e-4)            # 1e-5 px of coordinate rounding
    np.testing.assert_allclose(d.grad.cpu().numpy(), g["warp_ddepth"], rtol=2e-3, atol=5e-5)
    np.testing.assert_allclose(Tt.grad.cpu().numpy(), g["warp_dT"], rtol=2e-3, atol=5e-3)


def test_smooth_loss_vs_reference_golden(dev):
    g = load_golden("photo_reference.npz")
    tgt, src, depth, *_ = photo_case()
    disp = t((1.0 / depth).astype(np.float32), dev, True)
    sm = ph.get_smooth_loss(disp, t(tgt, dev))
    sm.backward()
    np.testing.assert_allclose(float(sm), float(g["smooth"]), rtol=2e-6)
    np.testing.assert_allclose(disp.grad.cpu().numpy(), g["smooth_ddisp"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("B,H,W", [(1, 2, 2), (2, 5, 7), (3, 32, 96)])
@pytest.mark.parametrize("use_ssim", [True, False])
def test_reprojection_loss_vs_oracle(dev, B, H, W, use_ssim):
    pred = synth.uniform((B, 3, H, W), "rp_p", 5, 0.0, 1.0).astype(np.float32)
    tgt = synth.uniform((B, 3, H, W), "rp_t", 5, 0.0, 1.0).astype(np.float32)
    w = synth.uniform((B, 1, H, W), "rp_w", 5, 0.0, 1.0).astype(np.float32)
    pc, tc = t(pred, None, True), t(tgt, None, True)
    ref = P.compute_reprojection_loss(pc, tc, use_ssim)
    (ref * t(w)).sum().backward()
    pg, tg = t(pred, dev, True), t(tgt, dev, True)
    out = ph.compute_reprojection_loss(pg, tg, use_ssim)
    (out * t(w, dev)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=5e-6)
    np.testing.assert_allclose(pg.grad.cpu().numpy(), pc.grad.numpy(), rtol=3e-4, atol=5e-5)
    np.testing.assert_allclose(tg.grad.cpu().numpy(), tc.grad.numpy(), rtol=3e-4, atol=5e-5)


def test_warp_border_clamp_and_identity(dev):
    """Identity pose (NOT the identity resampling: the reference normalises by W-1 and samples with align_corners=False,
    i.e. x -> x W/(W-1) - 0.5) against the oracle; then a large translation pushes every sample over the border, where
    grid_sample(padding_mode='border') clamps and the coordinate gradient is zero."""
    tgt, src, depth, K, inv_K, T = photo_case(B=2, H=16, W=24, seed=7)
    eye = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    out = ph.warp_frame(t(src, dev), t(depth, dev), t(K, dev), t(inv_K, dev), t(eye, dev))
    np.testing.assert_allclose(out.cpu().numpy(), P.warp_frame(t(src), t(depth), t(K), t(inv_K), t(eye)).numpy(), atol=2e-5)
    far = eye.copy()
    far[:, 0, 3] = 500.0                                        # every sample lands right of the frame
    d = t(depth, dev, True)
    Tt = t(far, dev, True)
    out = ph.warp_frame(t(src, dev), d, t(K, dev), t(inv_K, dev), Tt)
    ref = P.warp_frame(t(src), t(depth), t(K), t(inv_K), t(far))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.numpy(), atol=1e-5)
    out.sum().backward()
    assert float(d.grad.abs().max()) < 1e-6                     # x is clamped (zero gradient); v / w does not depend on the depth here


def test_photo_errors(dev):
    with pytest.raises(_lib.WmdError):
        ph.SSIM()(torch.zeros(1, 3, 4, 4), torch.zeros(1, 3, 4, 4))                      # CPU tensors: no fallback
    with pytest.raises(_lib.WmdError):
        ph.SSIM()(torch.zeros(1, 3, 1, 4, device=dev), torch.zeros(1, 3, 1, 4, device=dev))   # reflection padding needs H >= 2
    with pytest.raises(_lib.WmdError):
        ph.get_smooth_loss(torch.zeros(1, 2, 4, 4, device=dev), torch.zeros(1, 3, 4, 4, device=dev))


@pytest.mark.parametrize("hints", [False, True])
def test_trainer_loss_orchestration_vs_oracle(dev, hints):
    """generate_images_pred + compute_losses (KITTI/trainer.py:329-560) on the HIP operators vs the torch-CPU oracle:
    scalar losses, auto-mask agreement and gradients w.r.t. every disparity scale and the poses."""
    from util import loss_case
    inp, out = loss_case(hints=hints)
    opt = ph.LossOptions(height=32, width=64, frame_ids=[0, -1, 1] + (["s"] if hints else []), use_depth_hints=hints)
    grad_keys = [("disp", s) for s in range(4)] + [("cam_T_cam", 0, -1), ("cam_T_cam", 0, 1)]

    def run(device, mod):
        i2 = {k: torch.from_numpy(v).to(device) for k, v in inp.items()}
        o2 = {k: torch.from_numpy(v).to(device).requires_grad_(k in grad_keys) for k, v in out.items()}
        mod.generate_images_pred(i2, o2, opt)
        losses = mod.compute_losses(i2, o2, opt, tie_break_noise=0.0) if mod is ph else mod.compute_losses(i2, o2, opt)
        losses["loss"].backward()
        return losses, o2

    lg, og = run(dev, ph)
    lc, oc = run(torch.device("cpu"), P)
    for k in lc:
        np.testing.assert_allclose(float(lg[k]), float(lc[k]), rtol=2e-4, atol=1e-6, err_msg=k)
    for s in range(4):
        a, b = og["identity_selection/%d" % s].cpu(), oc["identity_selection/%d" % s]
        assert float((a != b).float().mean()) < 2e-3          # ties within rounding may fall either way
    for k in grad_keys:
        ga, gb = og[k].grad.cpu().numpy(), oc[k].grad.numpy()
        assert np.abs(ga - gb).max() <= 2e-3 * np.abs(gb).max() + 1e-7, k
