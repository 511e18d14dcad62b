"""Does a replayed hipGraph run two independent small kernels side by side?  Two under-filling convolutions (180
workgroups each on 256 CUs) captured (a) back to back on one stream, (b) forked onto two streams (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from wavelet_monodepth_amd import ops, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
xa = torch.randn(12, 128, 24, 80, device=dev)
xb = torch.randn(12, 128, 24, 80, device=dev)
wa = torch.randn(64, 128, 3, 3, device=dev) * 0.05
wb = torch.randn(64, 128, 3, 3, device=dev) * 0.05
ba = torch.zeros(64, device=dev)
fa = lambda: ops.conv2d_fused(xa, wa, ba, pad="reflect", act="elu")
fb = lambda: ops.conv2d_fused(xb, wb, ba, pad="reflect", act="elu")
with torch.no_grad():
    for _ in range(3):
        fa(); fb()
torch.cuda.synchronize()
REPS = 20


def capture(mode):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    keep = []
    with torch.no_grad(), torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for _ in range(REPS):
            if mode == "a":
                keep.append(fa())
            elif mode == "seq":
                keep.append(fa()); keep.append(fb())
            else:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    keep.append(fb())
                keep.append(fa())
                main.wait_stream(side)
    return g, keep


for mode in ("a", "seq", "fork"):
    g, keep = capture(mode)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-5s %.1f us per repetition" % (mode, e0.elapsed_time(e1) / (5 * REPS) * 1e3))

# eager launches on two streams (no graph)
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for mode in ("seq", "fork"):
    with torch.no_grad():
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            if mode == "seq":
                fa(); fb()
            else:
                with torch.cuda.stream(side):
                    fb()
                fa()
        main.wait_stream(side)
        e1.record(); torch.cuda.synchronize()
    print("eager %-5s %.1f us per repetition" % (mode, e0.elapsed_time(e1) / 100 * 1e3))

# two graphs replayed on two streams
ga, ka = capture("a")
gb = torch.cuda.CUDAGraph()
kb = []
with torch.no_grad(), torch.cuda.graph(gb):
    for _ in range(REPS):
        kb.append(fb())
torch.cuda.synchronize()
for mode in ("seq", "fork"):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        if mode == "seq":
            ga.replay(); gb.replay()
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                gb.replay()
            ga.replay()
            main.wait_stream(side)
    e1.record(); torch.cuda.synchronize()
    print("two graphs %-5s %.1f us per repetition" % (mode, e0.elapsed_time(e1) / (5 * REPS) * 1e3))
