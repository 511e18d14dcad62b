"""Development aid: launch times of the chained head GEMMs + completions at config 2 sizes (hipEvent pass)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import _lib, ops
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "12"))
for C, H, W in [(256, 12, 40), (128, 24, 80), (64, 48, 160)]:
    x = torch.randn(B, C, H, W, device=dev)
    yl = torch.randn(B, 1, H, W, device=dev)
    mk = lambda co=3, cm=None: [torch.randn(cm or C, C, 1, 1, device=dev) * 0.1, torch.randn(cm or C, device=dev), torch.randn(co, cm or C, 3, 3, device=dev) * 0.05, torch.randn(co, device=dev)]
    hp, hn = mk(), mk()
    ll = mk(1, C // 4) if C == 256 else None
    run = lambda: ops.head_fused_level_nograd(x, hp, hn, 2.0, None if ll else yl, 0.5, True, head_ll=ll, scale_ll=16.0)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(30):
        run()
    recs = _lib.profile_end()
    print("C=%d %dx%d B=%d: %s" % (C, H, W, B, ", ".join("%s %.1f us" % (r["kernel"], r["ms"] / r["calls"] * 1e3) for r in recs)))
