"""Host-side logic that needs no GPU: the sparse decoders' op model (the python-int `total_ops` the reference returns)
rebuilt from the REFERENCE's own masks and checked against the reference's own counts (tests/golden/kitti_sparse_*.npz),
and the pooled-buffer arithmetic."""
import numpy as np
import pytest

from util import R18, load_golden
from wavelet_monodepth_amd.kitti import sparse_decoder as SD

DEC = [16, 32, 64, 128, 256]


def _kitti_total_ops(gold, sparse_levels=(3, 2, 1)):
    static_ops, counters, counts = {}, [], {}
    for i in (4, 3, 2, 1):
        h, w = gold["lowres_mask|%d" % (i - 1)].shape[-2:]
        C = DEC[i]
        trunk = ((R18[4] if i == 4 else DEC[i + 1], C), (C + R18[i - 1], C))
        if i in sparse_levels:
            static_ops[i] = SD.level_static_ops(i, h, w, True)
            counters.append((i, None, trunk[0], trunk[1], (C, C), (C, 3)))
            counts[i] = tuple(int(gold["%s_mask|%d" % (m, i - 1)].sum()) for m in ("upconv0", "upconv1", "wavelet"))
        else:
            heads = ([(C, C // 4, C // 4, 1)] if i == 4 else []) + [(C, C, C, 3), (C, C, C, 3)]
            static_ops[i] = SD.level_static_ops(i, h, w, False, trunk, heads)
    return SD.resolve_total_ops(static_ops, counters, counts)


@pytest.mark.parametrize("name", ["64x64_thr-1", "64x64_thr0.01", "64x64_thr0.05", "64x64_thr0.1", "64x64_thr2",
                                  "96x160_thr0.15", "96x160_thr0.2"])
def test_kitti_sparse_op_model_from_reference_masks(name):
    gold = load_golden("kitti_sparse_r18_%s.npz" % name)
    per_scale, total = _kitti_total_ops(gold)
    for s in range(4):
        assert per_scale[s] == int(gold["total_ops|%d" % s]), "scale %d" % s
    assert total == int(gold["total_ops"])


def test_kitti_sparse_op_model_full_size_known_answer():
    """R18 640x192 with every pixel active: the notebook's printed 3.560 G."""
    gold = {}
    for i, (h, w) in zip((4, 3, 2, 1), ((6, 20), (12, 40), (24, 80), (48, 160))):
        gold["lowres_mask|%d" % (i - 1)] = np.ones((1, 1, h, w), np.uint8)
        gold["upconv0_mask|%d" % (i - 1)] = np.ones((1, 1, h, w), np.uint8)
        gold["upconv1_mask|%d" % (i - 1)] = np.ones((1, 1, 2 * h, 2 * w), np.uint8)
        gold["wavelet_mask|%d" % (i - 1)] = np.ones((1, 1, 2 * h, 2 * w), np.uint8)
    assert _kitti_total_ops(gold)[1] == 3560015775


def test_pool_rounding():
    assert [SD._round64(n) for n in (0, 1, 63, 64, 65, 128)] == [0, 64, 64, 64, 128, 128]
