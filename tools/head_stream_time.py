"""Development aid: launch time of the level-1 head kernel at config 2 / config 3 sizes (hipEvent pass, 30 launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import _lib, ops
dev = torch.device("cuda:0")
C = 32
for B, H, W in [(12, 96, 320), (8, 160, 512), (1, 96, 320)]:
    x = torch.randn(B, C, H, W, device=dev)
    yl = torch.randn(B, 1, H, W, device=dev)
    mk = lambda: [torch.randn(C, C, 1, 1, device=dev) * 0.2, torch.randn(C, device=dev), torch.randn(3, C, 3, 3, device=dev) * 0.1, torch.randn(3, device=dev)]
    hp, hn = mk(), mk()
    for _ in range(5):
        ops.head_fused_level_nograd(x, hp, hn, 2.0, yl, 0.5, True)
    torch.cuda.synchronize()
    _lib.profile_begin()
    for _ in range(30):
        ops.head_fused_level_nograd(x, hp, hn, 2.0, yl, 0.5, True)
    recs = _lib.profile_end()
    print("B=%d %dx%d dbg=%s th=%s: %s" % (B, H, W, os.environ.get("WMD_HS_DBG", "0"), os.environ.get("WMD_HEAD_STREAM_TH", "auto"),
                                     ", ".join("%s %.1f us" % (r["kernel"], r["ms"] / r["calls"] * 1e3) for r in recs)))
