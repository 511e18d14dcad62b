"""Launch times of the mask kernels of the sparse decoders (development aid): python tools/mask_microbench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import sparse_ops as S
dev = torch.device("cuda:0")
SPECS = [(1, 1), (1, 2), (2, 2), (2, 1), (2, 0)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def t(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for B in (1, 12):
    for (h, w) in ((12, 40), (24, 80), (48, 160)):
        yl = torch.randn(B, 1, 2 * h, 2 * w, device=dev)
        yh = torch.randn(B, 1, 3, h, w, device=dev)
        m = (torch.rand(B, h, w, device=dev) < 0.1).to(torch.uint8)
        cnt = torch.zeros(B, 3, device=dev, dtype=torch.int32)
        res = [B, h, w]
        res.append(t(lambda: S.mask_level(yl, yh, 0.5, SPECS)))
        res.append(t(lambda: S.mask_level(yl, yh, 0.5, SPECS, counts=(cnt, [1, 3, 4]))))
        res.append(t(lambda: S.mask_level(yl, yh, 0.5, SPECS[:1])))
        res.append(t(lambda: S.dilate_multi(m, SPECS)))
        res.append(t(lambda: S.dilate_multi(m, SPECS, counts=(cnt, [1, 3, 4]))))
        res.append(t(lambda: S.compact_multi(S.dilate_multi(m, SPECS)[2:])))
        print("B %2d %2dx%3d  mask_level %.1f  +counts %.1f  one-spec %.1f | dilate %.1f  +counts %.1f  dilate+compact3 %.1f us" % tuple(res))
