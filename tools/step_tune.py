"""In-step tuning of the trunk's (tile, split-K, finish) choices for bench.py's headline workload.

tuner.tune() times every candidate ALONE (eager launches, min of a few runs).  Inside the replayed forward a launch starts on
caches its predecessor left and ends into its successor's prologue, and the isolated winner is not always the in-step winner
(round 6: L14 <8,64,4,8> wins alone, <8,32,2,8> is 4 us per step faster inside the graph).  This tool starts from the committed
choices, takes the best few isolated candidates of every layer (a fresh sweep) and runs a coordinate descent on the time of the
WHOLE replayed step; it prints the resulting cache as JSON (last line) and writes it to --out.

    python tools/step_tune.py --out gpurun_out/step_tuned.json
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from wavelet_monodepth_amd import tuner  # noqa: E402


def step_ms(dec, feats, reps, steps):
    """median and minimum over `reps` loops of `steps` replays of a freshly captured forward"""
    dec.enable_graph(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.no_grad():
        for _ in range(4):
            dec(feats)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0.record()
            for _ in range(steps):
                dec(feats)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / steps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--cands", type=int, default=6)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--margin", type=float, default=0.0012, help="ms per step a candidate must gain to replace the incumbent")
    ap.add_argument("--workload", choices=["fwd", "fwd-1024"], default="fwd")
    ap.add_argument("--ksplits", default=None, help="comma-separated split counts to offer instead of tuner.KSPLITS")
    args = ap.parse_args()
    if args.ksplits:
        tuner.KSPLITS = tuple(int(v) for v in args.ksplits.split(","))
    dev = torch.device("cuda:0")
    committed = json.load(open(os.path.join(ROOT, "profiles", bench.TUNE_CACHE)))
    if args.workload == "fwd":
        dec, feats = bench.build_model(dev)
    else:       # bench.py's fwd_1024x320 extra: R50 channels, batch 8
        import numpy as np
        from wavelet_monodepth_amd import synth
        from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
        chans = [64, 256, 512, 1024, 2048]
        dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(chans)), seed=1).to(dev)
        feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(8, 320, 1024, chans, seed=1)]

    # 1. fresh isolated sweep of every layer -> tuner.ranked
    dec.enable_graph(False)
    with torch.no_grad():
        dec(feats)
    torch.cuda.synchronize()
    keys = [k for k in tuner.ranked if k in committed]
    print("layers:", len(keys), file=sys.stderr)

    # 2. start from the committed choices
    for k in keys:
        tuner._cache[k] = tuple(committed[k])
    base = step_ms(dec, feats, args.reps, args.steps)
    print("committed choices: %.4f ms (min %.4f)" % base, file=sys.stderr)
    best = base[0]
    for p in range(args.passes):
        changed = False
        for k in keys:
            inc = tuner._cache[k]
            cands = [(n, ks) for n, ks, _ in tuner.ranked[k][:args.cands] if (n, ks) != tuple(inc)]
            # the incumbent is re-measured with every layer: the reference drifts with the box's clocks
            tuner._cache[k] = inc
            ref = step_ms(dec, feats, args.reps, args.steps)[0]
            win, win_t = inc, ref
            for c in cands:
                tuner._cache[k] = c
                t = step_ms(dec, feats, args.reps, args.steps)[0]
                print("  pass %d %s %s %d: %.4f (incumbent %s %d %.4f)" % (p, k, c[0], c[1], t, inc[0], inc[1], ref), file=sys.stderr)
                if t < win_t - args.margin:
                    # confirm against the incumbent once more before switching
                    tuner._cache[k] = inc
                    ref2 = step_ms(dec, feats, args.reps, args.steps)[0]
                    tuner._cache[k] = c
                    t2 = step_ms(dec, feats, args.reps, args.steps)[0]
                    if t2 < ref2 - args.margin:
                        win, win_t = c, t2
            tuner._cache[k] = win
            if win != inc:
                changed = True
                print("pass %d: %s -> %s %d (%.4f -> %.4f)" % (p, k, win[0], win[1], ref, win_t), file=sys.stderr)
        if not changed:
            break
    final = step_ms(dec, feats, args.reps, args.steps)
    for k in keys:
        tuned = dict(tuner._cache)
    # A/B of the end points
    ab = []
    for _ in range(3):
        for k in keys:
            tuner._cache[k] = tuple(committed[k])
        a = step_ms(dec, feats, args.reps, args.steps)[0]
        for k in keys:
            tuner._cache[k] = tuned[k]
        b = step_ms(dec, feats, args.reps, args.steps)[0]
        ab.append((round(a, 4), round(b, 4)))
    print("committed vs step-tuned:", ab, "final %.4f" % final[0], file=sys.stderr)
    out = dict(committed)
    for k in keys:
        out[k] = list(tuned[k])
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=0)
    print(json.dumps({k: out[k] for k in keys}))


if __name__ == "__main__":
    main()
