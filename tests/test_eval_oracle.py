"""CPU: the evaluation oracle (oracle/eval_ref.py) against outputs of the reference's own compute_errors,
batch_post_process_disparity and compute_errors_nyu (tests/golden/eval_reference.npz, made by make_golden_eval.py)."""
import numpy as np

from oracle import eval_ref as E
from wavelet_monodepth_amd import synth
from util import load_golden


def u(tag, shape, lo, hi):
    return synth.uniform(shape, tag, 11, lo, hi).astype(np.float32)


def test_compute_errors_matches_reference_function():
    gold = load_golden("eval_reference.npz")
    for i, m in enumerate((1000, 46511, 7)):
        gt = u("ce_gt%d" % i, (m,), 1.0, 80.0)
        pred = gt * u("ce_ratio%d" % i, (m,), 0.5, 1.8)
        np.testing.assert_allclose(np.array(E.compute_errors(gt, pred), dtype=np.float64), gold["compute_errors_%d" % i], rtol=1e-7)


def test_post_process_matches_reference_function():
    gold = load_golden("eval_reference.npz")
    for i, (b, h, w) in enumerate(((2, 6, 40), (1, 5, 33), (3, 8, 2))):
        l = u("pp_l%d" % i, (b, h, w), 0.01, 1.0)
        r_raw = u("pp_r%d" % i, (b, h, w), 0.01, 1.0)
        np.testing.assert_allclose(E.batch_post_process_disparity(l, r_raw[:, :, ::-1]), gold["post_process_%d" % i], rtol=1e-12)


def test_compute_errors_nyu_matches_reference_function():
    gold = load_golden("eval_reference.npz")
    for i, shape in enumerate(((2, 50, 60), (1, 427, 561))):
        gt = u("nyu_gt%d" % i, shape, 0.5, 9.5)
        pred = np.clip(gt * u("nyu_ratio%d" % i, shape, 0.6, 1.6), 0.4, 10.0).astype(np.float32)
        # torch reduces in a different order than numpy: float32 rounding only
        np.testing.assert_allclose(np.array(E.compute_errors_nyu(pred, gt), dtype=np.float64), gold["compute_errors_nyu_%d" % i], rtol=2e-6)


def test_resize_is_identity_at_equal_size_and_exact_on_linear_ramps():
    a = u("rs", (7, 9), 0.0, 1.0)
    np.testing.assert_array_equal(E.resize_bilinear_cv2(a, 7, 9), a)
    ramp = np.tile(np.arange(8, dtype=np.float32), (4, 1))
    up = E.resize_bilinear_cv2(ramp, 8, 16)
    # interior samples of a linear ramp stay on the ramp: x_src = (x + 0.5) / 2 - 0.5
    np.testing.assert_allclose(up[0, 1:-1], (np.arange(1, 15) + 0.5) / 2 - 0.5, atol=1e-6)
    assert up[0, 0] == 0 and up[0, -1] == 7          # edge clamp


def test_kitti_chain_median_scaling_removes_global_scale():
    gt = u("kc_gt", (37, 122), 0.0, 90.0)
    gt[gt < 20] = 0                                   # sparse LiDAR-like validity
    disp = (1.0 / np.maximum(E.resize_bilinear_cv2(np.where(gt > 0, gt, 40.0).astype(np.float32), 12, 40), 1.0)).astype(np.float32)
    m1, r1, n1 = E.kitti_image_metrics(disp, gt)
    m2, r2, n2 = E.kitti_image_metrics(disp * 3.0, gt)
    assert n1 == n2 and n1 > 0
    np.testing.assert_allclose(r2 / r1, 3.0, rtol=1e-5)
    np.testing.assert_allclose(m1, m2, rtol=1e-4)
