"""Secondary workloads of BASELINE.json (configs[2..4]) — parity-tested elsewhere, timed here for the record:
  sparse   KITTI R18 640x192 sparse decoder, batch 1, threshold sweep, free-running and controlled-density masks
  nyu      NYUv2 DenseNet161-shaped DecoderWave 640x480, batch 4: forward and forward+backward
  r50      KITTI R50 1024x320 dense decoder, batch 8: forward
usage: python tools/config_bench.py [sparse] [nyu] [r50]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from wavelet_monodepth_amd import synth

dev = torch.device("cuda:0")
which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["sparse", "nyu", "r50"]


def timeit(fn, n=20, warm=3):
    # (the cyclic garbage collector is parked for the timed loop: a generation-2 pass over the captured graphs' tensors is a
    # 20-30 ms host pause that now and then landed inside one of these 15 ms loops and tripled that line)
    import gc
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    finally:
        gc.enable()


if "sparse" in which:
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder
    chans = [64, 64, 128, 256, 512]
    sp = synth.fill_state_dict(SparseDepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
    dn = DepthWaveProgressiveDecoder(np.array(chans)).to(dev)
    dn.load_state_dict(sp.state_dict())
    feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(1, 192, 640, chans, seed=1)]
    with torch.no_grad():
        t_dense = timeit(lambda: dn(feats))
    print("sparse: dense decoder batch 1: %.3f ms (%.0f frames/s)" % (t_dense * 1e3, 1 / t_dense))
    dn.enable_graph(True)
    with torch.no_grad():
        t_dense_g = timeit(lambda: dn(feats))
    print("sparse: dense decoder batch 1, hipGraph replay: %.3f ms (%.0f frames/s)" % (t_dense_g * 1e3, 1 / t_dense_g))
    sp.enable_graph("--no-graph" not in sys.argv)
    for thr in (-1.0, 0.01, 0.02, 0.05, 0.1, 0.15, 0.2):
        out = sp(feats, thr)
        dens = [float(out[("wavelet_mask", s)].float().mean()) for s in (2, 1, 0)]
        t = timeit(lambda: sp(feats, thr), n=10)
        print("sparse: thresh %5.2f  density(scale2,1,0) = %.2f %.2f %.2f  total_ops %.3f G  %.3f ms  %.0f frames/s" % (
            thr, dens[0], dens[1], dens[2], out["total_ops"] / 1e9, t * 1e3, 1 / t))
    # controlled density: random seed pixels dilated into blobs covering ~p of each level's coarse grid
    for p in (0.05, 0.10, 0.20, 0.50):
        force = {}
        for i, (h, w) in zip((3, 2, 1), ((12, 40), (24, 80), (48, 160))):
            m = (torch.from_numpy(synth.uniform((h, w), "dens%d" % i, 3, 0.0, 1.0)) < p).to(torch.uint8).to(dev)
            force[i] = m
        out = sp(feats, 0.05, _force_masks=force)
        dens = [float(out[("wavelet_mask", s)].float().mean()) for s in (2, 1, 0)]
        t = timeit(lambda: sp(feats, 0.05, _force_masks=force), n=10)
        print("sparse: injected density %.2f -> masks %.2f %.2f %.2f  total_ops %.3f G  %.3f ms  %.0f frames/s" % (
            p, dens[0], dens[1], dens[2], out["total_ops"] / 1e9, t * 1e3, 1 / t))
    # contour masks: thin outlines of a smooth field (what depth discontinuities look like), per-level densities
    for ps in ((0.10, 0.10, 0.10), (0.30, 0.10, 0.03), (0.10, 0.03, 0.01)):
        force = {i: torch.from_numpy(synth.contour_mask(h, w, p, "contour", 3)).to(dev)
                 for (i, (h, w)), p in zip(((3, (12, 40)), (2, (24, 80)), (1, (48, 160))), ps)}
        out = sp(feats, 0.05, _force_masks=force)
        dens = [float(out[("wavelet_mask", s)].float().mean()) for s in (2, 1, 0)]
        t = timeit(lambda: sp(feats, 0.05, _force_masks=force), n=10)
        print("sparse: contour masks, densities %.2f %.2f %.2f  total_ops %.3f G  %.3f ms  %.0f frames/s" % (
            dens[0], dens[1], dens[2], out["total_ops"] / 1e9, t * 1e3, 1 / t))

if "sparse-throughput" in which:
    # BASELINE config 4 as a THROUGHPUT question: a single 640x192 frame cannot fill 256 CUs (dense: ~25 dependent launches,
    # sparse: ~35, both latency-bound), so independent frames are decoded concurrently -- K captured graphs replayed on K
    # streams (the sparse forward never waits for the host: total_ops is lazy) -- against the dense decoder run the same
    # way and as one batch of K.
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder
    chans = [64, 64, 128, 256, 512]
    K = 12
    sp = synth.fill_state_dict(SparseDepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
    dn = DepthWaveProgressiveDecoder(np.array(chans)).to(dev)
    dn.load_state_dict(sp.state_dict())
    dn.two_stream_graphs = False
    sp._graphs._max = dn._graphs._max = 4 * K
    frames = [[torch.from_numpy(f).to(dev) for f in synth.encoder_features(1, 192, 640, chans, seed=10 + k)] for k in range(K)]
    batch = [torch.cat([fr[j] for fr in frames], 0) for j in range(5)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    sp.enable_graph(True)
    dn.enable_graph(True)

    def concurrent(fn, n=20):
        def once():
            cur = torch.cuda.current_stream()
            for k in range(K):
                streams[k].wait_stream(cur)
                with torch.cuda.stream(streams[k]):
                    fn(k)
            for k in range(K):
                cur.wait_stream(streams[k])
        t = timeit(once, n=n)
        return t

    with torch.no_grad():
        t_b = timeit(lambda: dn(batch), n=20)
        print("throughput: dense decoder, one batch of %d: %.3f ms  %.0f frames/s" % (K, t_b * 1e3, K / t_b))
        t_d = concurrent(lambda k: dn(frames[k]))
        print("throughput: dense decoder, %d frames on %d streams: %.3f ms  %.0f frames/s" % (K, K, t_d * 1e3, K / t_d))
    # batched sparse decode: the K frames, each with its own masks, through ONE chain of launches
    with torch.no_grad():
        for p in (0.05, 0.10, 0.20, 0.50, 1.00):
            fb = {i: (torch.from_numpy(synth.uniform((K, h, w), "bdens%d" % i, 3, 0.0, 1.0)) < p).to(torch.uint8).to(dev)
                  for i, (h, w) in zip((3, 2, 1), ((12, 40), (24, 80), (48, 160)))}
            out = sp(batch, 0.05, _force_masks=fb)
            t_s = timeit(lambda: sp(batch, 0.05, _force_masks=fb), n=20)
            print("throughput: sparse decoder, ONE batch of %d, injected density %.2f (mean total_ops %.3f G): %.3f ms  %.0f frames/s"
                  % (K, p, float(np.mean(out["total_ops"])) / 1e9, t_s * 1e3, K / t_s))
        for ps in ((0.10, 0.10, 0.10), (0.30, 0.10, 0.03), (0.10, 0.03, 0.01)):
            fb = {i: torch.from_numpy(np.stack([synth.contour_mask(h, w, p, "contour", 10 + k) for k in range(K)])).to(dev)
                  for (i, (h, w)), p in zip(((3, (12, 40)), (2, (24, 80)), (1, (48, 160))), ps)}
            out = sp(batch, 0.05, _force_masks=fb)
            t_s = timeit(lambda: sp(batch, 0.05, _force_masks=fb), n=20)
            print("throughput: sparse decoder, ONE batch of %d, contour masks of density %.2f %.2f %.2f (mean total_ops %.3f G): %.3f ms  %.0f frames/s"
                  % (K, ps[0], ps[1], ps[2], float(np.mean(out["total_ops"])) / 1e9, t_s * 1e3, K / t_s))
        for thr in (0.05, 0.10, 0.15, 0.20):
            out = sp(batch, thr)
            t_s = timeit(lambda: sp(batch, thr), n=20)
            dens = [float(out[("wavelet_mask", s)].float().mean()) for s in (2, 1, 0)]
            print("throughput: sparse decoder, ONE batch of %d, thresh %.2f (densities %.2f %.2f %.2f, mean total_ops %.3f G): %.3f ms  %.0f frames/s"
                  % (K, thr, dens[0], dens[1], dens[2], float(np.mean(out["total_ops"])) / 1e9, t_s * 1e3, K / t_s))
    for p in (0.10,):
        forces = []
        for k in range(K):
            forces.append({i: (torch.from_numpy(synth.uniform((h, w), "dens%d_%d" % (i, k), 3, 0.0, 1.0)) < p).to(torch.uint8).to(dev)
                           for i, (h, w) in zip((3, 2, 1), ((12, 40), (24, 80), (48, 160)))})
        t_s = concurrent(lambda k: sp(frames[k], 0.05, _force_masks=forces[k]))
        out = sp(frames[0], 0.05, _force_masks=forces[0])
        print("throughput: sparse decoder, injected density %.2f (total_ops %.3f G), %d frames on %d streams: %.3f ms  %.0f frames/s"
              % (p, out["total_ops"] / 1e9, K, K, t_s * 1e3, K / t_s))
    for thr in (0.15,):
        t_s = concurrent(lambda k: sp(frames[k], thr))
        out = sp(frames[0], thr)
        dens = [float(out[("wavelet_mask", s)].float().mean()) for s in (2, 1, 0)]
        print("throughput: sparse decoder, thresh %.2f (densities %.2f %.2f %.2f, total_ops %.3f G), %d frames on %d streams: %.3f ms  %.0f frames/s"
              % (thr, dens[0], dens[1], dens[2], out["total_ops"] / 1e9, K, K, t_s * 1e3, K / t_s))

if "nyu" in which:
    from wavelet_monodepth_amd.nyu import DecoderWave
    enc = [96, 96, 192, 384, 2208]
    B = 4
    dec = synth.fill_state_dict(DecoderWave(enc_features=enc), seed=9).to(dev)
    feats = [torch.from_numpy(synth.normal((B, c, 480 >> (k + 1), 640 >> (k + 1)), "nyu%d" % k, 9)).to(dev) for k, c in enumerate(enc)]
    with torch.no_grad():
        t = timeit(lambda: dec(feats), n=10)
    gmac = 33.325
    print("nyu: DecoderWave 640x480 batch %d forward: %.3f ms  %.1f frames/s  %.1f TFLOP/s" % (B, t * 1e3, B / t, 2 * gmac * B / t / 1e3))
    dec.enable_graph(True)
    with torch.no_grad():
        t = timeit(lambda: dec(feats), n=20)
    dec.enable_graph(False)
    print("nyu: DecoderWave 640x480 batch %d forward, hipGraph replay: %.3f ms  %.1f frames/s  %.1f TFLOP/s" % (B, t * 1e3, B / t, 2 * gmac * B / t / 1e3))
    fg = [f.clone().requires_grad_(True) for f in feats]

    def step():
        out = dec(fg)
        sum(out[("disp", s)].mean() for s in range(4)).backward()
    t = timeit(step, n=5, warm=2)
    print("nyu: DecoderWave 640x480 batch %d forward+backward: %.3f ms  %.1f frames/s" % (B, t * 1e3, B / t))

if "r50" in which:
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    chans = [64, 256, 512, 1024, 2048]
    B = 8
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
    feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(B, 320, 1024, chans, seed=1)]
    with torch.no_grad():
        t = timeit(lambda: dec(feats), n=10)
    print("r50: dense decoder 1024x320 batch %d forward: %.3f ms  %.1f frames/s  %.1f TFLOP/s" % (B, t * 1e3, B / t, 2 * 17.19 * B / t / 1e3))
    for two in (False, True):
        dec.two_stream_graphs = two
        dec.enable_graph(True)
        with torch.no_grad():
            t = timeit(lambda: dec(feats), n=20)
        print("r50: dense decoder 1024x320 batch %d forward, hipGraph replay (%s): %.3f ms  %.1f frames/s  %.1f TFLOP/s" % (
            B, "trunk/heads on two streams" if two else "one graph", t * 1e3, B / t, 2 * 17.19 * B / t / 1e3))
    dec.enable_graph(False)
