"""Development aid (round 6): head_stream_kernel vs head_level_kernel -- same inputs, max abs difference per output and the
launch time of each (hipEvent pass).  The old kernel is reached with WMD_HEAD_STREAM=0 in a child process (the switch is read
once per process)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

SHAPES = [(12, 96, 320), (8, 160, 512), (2, 12, 40), (2, 5, 7), (2, 10, 84), (2, 2, 2), (1, 96, 320), (3, 33, 65), (2, 50, 31)]


def run(tag):
    from wavelet_monodepth_amd import _lib, ops
    dev = torch.device("cuda:0")
    C = 32
    res = {}
    for B, H, W in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(B * 1000 + H * 10 + W)
        rn = lambda *s: torch.randn(*s, generator=g)
        x = rn(B, C, H, W).to(dev)
        yl = (rn(B, 1, H, W) * 2 + 4).to(dev)
        mk = lambda: [(rn(C, C, 1, 1) * 0.2).to(dev), rn(C).to(dev), (rn(3, C, 3, 3) * 0.1).to(dev), rn(3).to(dev)]
        hp, hn = mk(), mk()
        for _ in range(2):
            out = ops.head_fused_level_nograd(x, hp, hn, 2.0, yl, 0.5, True)
        torch.cuda.synchronize()
        _lib.profile_begin()
        for _ in range(20):
            out = ops.head_fused_level_nograd(x, hp, hn, 2.0, yl, 0.5, True)
        recs = _lib.profile_end()
        res[(B, H, W)] = ([o.cpu() for o in out[:3]], {r["kernel"]: r["ms"] / r["calls"] * 1e3 for r in recs})
    torch.save(res, "/tmp/hs_%s.pt" % tag)
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
        sys.exit(0)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    subprocess.check_call([sys.executable, __file__, "old"], env=dict(os.environ, WMD_HEAD_STREAM="0"))
    new = run("new")
    old = torch.load("/tmp/hs_old.pt")
    for k in SHAPES:
        d = [float((a - b).abs().max()) for a, b in zip(new[k][0], old[k][0])]
        fin = all(bool(torch.isfinite(a).all()) for a in new[k][0])
        print("B=%d %dx%d: max|new-old| yh %.2e out %.2e disp %.2e finite=%s | new %s | old %s" % (
            k + tuple(d) + (fin, new[k][1], old[k][1])))
