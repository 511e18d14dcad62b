"""GPU: evaluation arithmetic (csrc/wmd_eval.hip through the C ABI) against the CPU oracle and the goldens produced by
the reference's own functions."""
import numpy as np
import pytest
import torch

from oracle import eval_ref as E
from wavelet_monodepth_amd import _lib, evaluation as ev, synth
from util import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


def u(tag, shape, lo, hi):
    return synth.uniform(shape, tag, 11, lo, hi).astype(np.float32)


def g(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_compute_errors_vs_reference_golden(dev):
    gold = load_golden("eval_reference.npz")
    for i, m in enumerate((1000, 46511, 7)):
        gt = u("ce_gt%d" % i, (m,), 1.0, 80.0)
        pred = gt * u("ce_ratio%d" % i, (m,), 0.5, 1.8)
        out = ev.compute_errors(g(gt[None], dev), g(pred[None], dev))[0].cpu().numpy()
        np.testing.assert_allclose(out, gold["compute_errors_%d" % i], rtol=1e-5)   # fp64 sums vs numpy's fp32 pairwise sums


def test_compute_errors_nyu_vs_reference_golden(dev):
    gold = load_golden("eval_reference.npz")
    for i, shape in enumerate(((2, 50, 60), (1, 427, 561))):
        gt = u("nyu_gt%d" % i, shape, 0.5, 9.5)
        pred = np.clip(gt * u("nyu_ratio%d" % i, shape, 0.6, 1.6), 0.4, 10.0).astype(np.float32)
        # the reference pools all images of the call into one mean: do the same by flattening
        out = ev.compute_errors_nyu(g(pred, dev), g(gt, dev)).cpu().numpy()          # reference semantics: one global reduction
        rows = ev.compute_errors_nyu(g(pred, dev), g(gt, dev), per_image=True)
        assert rows.shape == (pred.shape[0], 6)
        np.testing.assert_allclose(rows[:, 0].mean().item(), out[0], rtol=1e-5)      # abs_rel is a plain mean: rows average to it
        np.testing.assert_allclose(out, gold["compute_errors_nyu_%d" % i], rtol=1e-5)


def test_flip_postprocess_vs_reference_golden(dev):
    gold = load_golden("eval_reference.npz")
    for i, (b, h, w) in enumerate(((2, 6, 40), (1, 5, 33), (3, 8, 2))):
        l = u("pp_l%d" % i, (b, h, w), 0.01, 1.0)
        r_raw = u("pp_r%d" % i, (b, h, w), 0.01, 1.0)
        out = ev.flip_postprocess(g(l, dev), g(r_raw, dev)).cpu().numpy()
        np.testing.assert_allclose(out, gold["post_process_%d" % i], rtol=2e-7, atol=1e-7)


def _kitti_case(B, h, w, H, W, tag):
    gt = u(tag + "gt", (B, H, W), 0.0, 95.0)
    gt[u(tag + "hole", (B, H, W), 0.0, 1.0) < 0.7] = 0          # ~30 % valid, like projected LiDAR
    disp = u(tag + "disp", (B, h, w), 0.01, 0.3)
    return disp, gt


@pytest.mark.parametrize("B,h,w,H,W", [(3, 12, 40, 37, 122), (2, 48, 160, 93, 310), (1, 192, 640, 375, 1242)])
@pytest.mark.parametrize("mode", ["eigen_median", "eigen_stereo", "benchmark_median"])
def test_kitti_chain_vs_oracle(dev, B, h, w, H, W, mode):
    disp, gt = _kitti_case(B, h, w, H, W, "k%d%d" % (H, W))
    eigen = mode.startswith("eigen")
    stereo = mode.endswith("stereo")
    err, ratios, nvalid = ev.kitti_metrics(g(disp, dev), g(gt, dev), eval_split="eigen" if eigen else "benchmark", eval_stereo=stereo)
    err, nvalid = err.cpu().numpy(), nvalid.cpu().numpy()
    for b in range(B):
        ref, ratio, n = E.kitti_image_metrics(disp[b], gt[b], eigen=eigen, pred_depth_scale_factor=5.4 if stereo else 1.0,
                                              disable_median_scaling=stereo)
        assert nvalid[b] == n                                                    # mask: exact
        if not stereo:
            np.testing.assert_allclose(float(ratios[b]), ratio, rtol=2e-6)       # np.median: order statistics are exact,
        np.testing.assert_allclose(err[b], np.array(ref, dtype=np.float64), rtol=2e-5, atol=1e-7)   # resampling rounding only


def test_kitti_chain_median_is_exact_order_statistic(dev):
    """Radix select == np.median, bit for bit, for odd and even counts and with ties."""
    for n_valid in (1, 2, 5, 64, 1001):
        H, W = 40, 50
        gt = np.zeros((1, H, W), np.float32)
        vals = u("med%d" % n_valid, (n_valid,), 1.0, 70.0)
        vals[: n_valid // 3] = vals[0]                                           # ties
        gt.reshape(-1)[:n_valid] = vals
        disp = np.full((1, H, W), 0.1, np.float32)                               # same size: the resize is the identity
        _, ratios, nv = ev.kitti_metrics(g(disp, dev), g(gt, dev), eval_split="benchmark")
        assert int(nv[0]) == n_valid
        assert float(ratios[0]) == np.float32(np.median(vals)) / np.float32(np.median(np.full(n_valid, np.float32(1.0) / np.float32(0.1))))


def test_empty_mask_and_errors(dev):
    gt = np.zeros((1, 8, 8), np.float32)
    err, ratios, nv = ev.kitti_metrics(g(np.full((1, 4, 4), 0.1, np.float32), dev), g(gt, dev), eval_split="benchmark")
    assert int(nv[0]) == 0 and torch.isnan(err).all()
    with pytest.raises(_lib.WmdError):
        ev.kitti_metrics(torch.zeros(1, 4, 4), torch.zeros(1, 8, 8))             # CPU tensors: no fallback
    with pytest.raises(_lib.WmdError):
        ev.kitti_metrics(torch.zeros(2, 4, 4, device=dev), torch.zeros(1, 8, 8, device=dev))


def test_nyu_prediction_chain_vs_oracle(dev):
    pred = u("nyup", (2, 1, 240, 320), 0.3, 11.0)
    crop = (20, 459, 24, 615)                                                    # NYUv2/evaluate.py Eigen crop
    out = ev.nyu_prediction(g(pred, dev), crop).cpu().numpy()
    ref = E.nyu_prediction_chain(pred, crop)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out, ref, rtol=2e-6, atol=2e-6)


def test_abs_rel_of_hip_decoder_vs_oracle_decoder(dev):
    """BASELINE.json's secondary metric: depth from the HIP decoder vs depth from the CPU oracle decoder, scored with
    the evaluation chain above, must agree to far better than 0.001 abs_rel."""
    from oracle import decoder_ref as R
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    from util import R18, kitti_feats
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=4)
    feats = kitti_feats(2, 64, 128)
    sd = {k: v.detach() for k, v in dec.state_dict().items()}
    with torch.no_grad():
        ref = R.kitti_wave_decoder(feats, sd)[("disp", 0)][:, 0]
        out = dec.to(dev)([f.to(dev) for f in feats])[("disp", 0)][:, 0]
    min_disp, max_disp = 1 / 100.0, 1 / 0.1                                      # disp_to_depth, layers.py:16-25
    depth_ref = 1 / (min_disp + (max_disp - min_disp) * ref)
    pred_disp = (min_disp + (max_disp - min_disp) * out).contiguous()
    err, _, _ = ev.kitti_metrics(pred_disp, depth_ref.to(dev).contiguous(), eval_split="benchmark", disable_median_scaling=True,
                                 min_depth=1e-3, max_depth=200.0)
    assert float(err[:, 0].max()) < 1e-4, err
