"""Round 6 probe: how long do the three chained head GEMM launches (levels 4 / 3 / 2 of config 2, batch 12) take back to back on one
stream, and launched together on three streams (= what ONE merged launch could reach: the levels are too small to balance the 256 CUs
alone, profiles/r06_notes.md section 6)?  And the completion launches behind them: per level vs the chained one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import _lib, ops
dev = torch.device("cuda:0")
B = 12
lv = []
for C, H, W in [(256, 12, 40), (128, 24, 80), (64, 48, 160)]:
    x = torch.randn(B, C, H, W, device=dev)
    mk = lambda co=3, cm=None: [torch.randn(cm or C, C, 1, 1, device=dev) * 0.1, torch.randn(cm or C, device=dev), torch.randn(co, cm or C, 3, 3, device=dev) * 0.05, torch.randn(co, device=dev)]
    lv.append((x, mk(), mk(), mk(1, C // 4) if C == 256 else None))
streams = [torch.cuda.Stream() for _ in lv]
def serial():
    return [ops.head_fused_gemm_nograd(x, hp, hn, ll) for x, hp, hn, ll in lv]
def parallel():
    cur = torch.cuda.current_stream()
    out = []
    for s, (x, hp, hn, ll) in zip(streams, lv):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            out.append(ops.head_fused_gemm_nograd(x, hp, hn, ll))
    for s in streams:
        cur.wait_stream(s)
    return out
def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
with torch.no_grad():
    print("three chained-GEMM launches, one stream: %.1f us (best of 30, incl. launch gaps)" % timeit(serial))
    print("three chained-GEMM launches, three streams: %.1f us" % timeit(parallel))
    items = serial()
    sc = [2.0 ** (k - 1) for k in (4, 3, 2)]; ds = [1.0 / 2 ** (k - 1) for k in (4, 3, 2)]
    print("chained completion of levels 4-2 (one launch): %.1f us" % timeit(lambda: ops.head_shiftsum_chain_nograd(items, sc, ds, scale_ll=16.0)))
