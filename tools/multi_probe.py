"""Round 6 probe: what each level costs inside / outside the merged first-stage launch (config 2, batch 12)."""
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
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(n): fn()
    recs = _lib.profile_end()
    return ", ".join("%s %.1f" % (r["kernel"], r["ms"] / r["calls"] * 1e3) for r in recs if "pack" not in r["kernel"])
with torch.no_grad():
    for name, sub in (("L4", [0]), ("L3", [1]), ("L2", [2]), ("L4+L3", [0, 1]), ("L3+L2", [1, 2]), ("all", [0, 1, 2])):
        if len(sub) == 1:
            x, hp, hn, ll = lv[sub[0]]
            print(name, "own launch:", t(lambda: ops.head_fused_gemm_nograd(x, hp, hn, ll)))
        if len(sub) >= 2:
            print(name, "merged:", t(lambda: ops.head_fused_gemm_multi_nograd([lv[i] for i in sub])))
