"""Round 6 probe: completions of levels 4 / 3 / 2 at batch 12 behind the merged first-stage launch: three per-level launches vs
chained(4,3) + level 2 vs chained(4,3,2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import _lib, ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "12"))
lv = []
for C, H, W in [(256, 12, 40), (128, 24, 80), (64, 48, 160)]:
    x = torch.randn(B, C, H, W, device=dev)
    mk = lambda co=3, cm=None: [torch.randn(cm or C, C, 1, 1, device=dev) * 0.1, torch.randn(cm or C, device=dev), torch.randn(co, cm or C, 3, 3, device=dev) * 0.05, torch.randn(co, device=dev)]
    lv.append((x, mk(), mk(), mk(1, C // 4) if C == 256 else None))
sc = [2.0 ** (k - 1) for k in (4, 3, 2)]; ds = [1.0 / 2 ** (k - 1) for k in (4, 3, 2)]
def per_level(items):
    yl = None
    for k, it in enumerate(items):
        yh, out, disp, yl_ll = ops.head_shiftsum_item_nograd(it, sc[k], ds[k], yl=yl, scale_ll=16.0)
        yl = out
    return yl
def two_plus_one(items):
    r = ops.head_shiftsum_chain_nograd(items[:2], sc[:2], ds[:2], scale_ll=16.0)
    return ops.head_shiftsum_item_nograd(items[2], sc[2], ds[2], yl=r[1][1], scale_ll=16.0)[1]
def all_three(items):
    return ops.head_shiftsum_chain_nograd(items, sc, ds, scale_ll=16.0)[2][1]
def timeit(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = []
    for _ in range(n + 5):
        items = ops.head_fused_gemm_multi_nograd(lv)     # the planes are fresh in the caches, as in the decoder
        e0.record(); fn(items); e1.record(); e1.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3)
    best = sorted(best[5:])
    return best[0], best[len(best) // 2]
with torch.no_grad():
    a = per_level(ops.head_fused_gemm_multi_nograd(lv)); b = two_plus_one(ops.head_fused_gemm_multi_nograd(lv)); c = all_three(ops.head_fused_gemm_multi_nograd(lv))
    print("equal:", torch.equal(a, b), torch.equal(a, c))
    for name, fn in (("three per-level launches", per_level), ("chained(4,3) + level 2", two_plus_one), ("chained(4,3,2)", all_three)):
        print("B=%d %s: best %.1f us, median %.1f us" % ((B, name) + timeit(fn)))
