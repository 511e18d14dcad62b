"""Config 2 (batch 12) as ONE graph vs two half-batch graphs replayed on two streams (development aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from wavelet_monodepth_amd import synth, tuner
from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder

dev = torch.device("cuda:0")
tuner.preload(os.path.join(ROOT, "profiles", "r01_tune_cache_config2.json"))
chans = [64, 64, 128, 256, 512]
dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(chans)), seed=0).to(dev)
feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(12, 192, 640, chans, seed=0)]
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
two = os.environ.get("WMD_TWO_STREAM_GRAPHS", "1") == "1"
dec.enable_graph(True)
n = 12 // parts
halves = [[f[k * n:(k + 1) * n] for f in feats] for k in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts - 1)]
main = torch.cuda.current_stream()


def full():
    return dec(feats)


def split():
    outs = []
    for k in range(1, parts):
        streams[k - 1].wait_stream(main)
        with torch.cuda.stream(streams[k - 1]):
            outs.append(dec(halves[k]))
    outs.append(dec(halves[0]))
    for s in streams:
        main.wait_stream(s)
    return outs


def timeit(fn, n=100, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    print("two_stream_graphs=%d  full batch 12: %.4f ms" % (two, timeit(full)))
    print("two_stream_graphs=%d  %d x batch %d on %d streams: %.4f ms" % (two, parts, n, parts, timeit(split)))
