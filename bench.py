#!/usr/bin/env python
"""Headline benchmark: decoder+IDWT forward frames/sec @640x192, batch 12 per GPU (BASELINE.json).

A "step" = one forward of DepthWaveProgressiveDecoder (ResNet18 channels) over one synthetic batch of
encoder features already resident in HBM: 8 fused 3x3 MFMA convolutions, 9 wavelet heads, 4 Haar IDWTs
-> 4 disparity maps + 16 coefficient planes.  N GPUs = N independent shards of the batch dimension
(weak scaling; the forward path has no exchange step, so no collective is issued).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

R18 = [64, 64, 128, 256, 512]
HEIGHT, WIDTH, BATCH = 192, 640, 12
FLOP_PER_FRAME = 6.947e9     # conv MACs x2, SURVEY.md §8(d) / BASELINE.md §2
PEAK_F32_MFMA = 157.3        # TFLOP/s, MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)


def build_model(dev):
    from wavelet_monodepth_amd import synth
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder

    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev)
    dec.enable_graph(os.environ.get("WMD_BENCH_GRAPH", "1") != "0")
    feats = [torch.from_numpy(f).to(dev) for f in synth.encoder_features(BATCH, HEIGHT, WIDTH, R18, seed=1)]
    return dec, feats


def cpu_baseline(dec, feats, budget_s=20.0):
    """The CPU oracle (PyTorch-CPU/oneDNN restatement of the reference decoder) on the host cores,
    on a bounded sample of the same workload.  oneDNN does not scale to every hardware thread of a big
    host, so one full-batch pass per candidate thread count picks the best first (the baseline gets its
    best case), then whole 12-frame batches are timed until ~budget_s elapsed."""
    from oracle import decoder_ref as R

    sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    cf = [f.cpu() for f in feats]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # oneDNN stops scaling (and then collapses) long before a 256-thread host is full, and what is best for a
    # 2-frame batch is not best for 12 frames: the thread count is chosen on the workload itself
    cands = sorted({c for c in (64, 32, 16, 8, min(avail, 4)) if 1 <= c <= avail}, reverse=True)
    best, best_t = cands[-1], float("inf")
    probe_fps = {}
    with torch.no_grad():
        for c in cands:
            torch.set_num_threads(c)
            R.kitti_wave_decoder(cf, sd)
            t0 = time.perf_counter()
            R.kitti_wave_decoder(cf, sd)
            dt = time.perf_counter() - t0
            probe_fps[str(c)] = round(BATCH / dt, 1)
            if dt < best_t:
                best, best_t = c, dt
        torch.set_num_threads(best)
        R.kitti_wave_decoder(cf, sd)  # warm-up
        times = []
        t_start = time.perf_counter()
        while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 30):
            t0 = time.perf_counter()
            R.kitti_wave_decoder(cf, sd)
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(BATCH / med, 2), "unit": "frames/s", "cores": best, "kind": "port",
            "probe_frames_per_s_by_threads": probe_fps,   # one full batch each; 8 threads is SURVEY.md's probe setting
            "sample": "%d timed passes of the same 12x640x192 batch (median %.3f s/pass) after 1 warm-up; torch %s CPU; "
                      "%d threads = best of %s on one full-batch pass each; host exposes %d hardware threads"
                      % (len(times), med, torch.__version__, best, cands, avail)}


def train_main(args, rank, local_rank, world, dev):
    """Secondary workload (BASELINE.json configs[2]): KITTI ResNet50 1024x320 training step, batch 8 per GPU —
    torch encoder + HIP decoder forward/backward, gradient all-reduce through RCCL (decoder bucket first, on a
    side stream), Adam.  Reported for scaling studies; the headline metric stays the forward workload."""
    import torch.distributed as dist
    from wavelet_monodepth_amd import synth
    from wavelet_monodepth_amd.ddp import GradientExchange, monodepth_groups
    from wavelet_monodepth_amd.encoders import ResnetEncoder
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder

    H, W, B = args.height, args.width, args.batch
    torch.manual_seed(0)
    enc = ResnetEncoder(args.num_layers).to(dev)
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(enc.num_ch_enc), seed=1).to(dev)
    params = list(enc.parameters()) + list(dec.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-5)
    gx = GradientExchange(monodepth_groups(enc, dec), world=world, rank=rank, backend="rccl" if world > 1 else "torch") \
        if world > 1 else None
    img = torch.from_numpy(synth.uniform((B, 3, H, W), "img%d" % rank, 0, 0.0, 1.0)).to(dev)
    target = [torch.from_numpy(synth.uniform((B, 1, H >> s, W >> s), "tgt%d" % s, rank, 0.05, 0.95)).to(dev) for s in range(4)]

    def step():
        if gx is not None:
            gx.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        out = dec(enc(img))
        loss = sum((out[("disp", s)] - target[s]).abs().mean() for s in range(4))
        loss.backward()
        if gx is not None:
            gx.finish()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(loss)
    if rank == 0:
        print(json.dumps({
            "metric": "training frames/sec (encoder + wavelet decoder fwd+bwd + Adam) @%dx%d bs%d/GPU" % (W, H, B),
            "value": round(B * args.steps * world / elapsed, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "KITTI ResNet%d %dx%d training step, batch %d per GPU, loss = sum_s mean|disp_s - target_s| "
                                   "(BASELINE.json configs[2])" % (args.num_layers, W, H, B),
                       "global_batch": B * world, "parallelism": "dp%d, RCCL bucketed all-reduce" % world},
            "gradient_bytes": None if gx is None else gx.message_bytes()}))
    if gx is not None:
        gx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["fwd", "train"], default="fwd")
    ap.add_argument("--num-layers", type=int, default=50)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    # WMD_BENCH_BACKEND=gloo + WMD_BENCH_SHARE_DEVICES=1: rehearsal of the N>1 launch contract on a box with fewer GPUs
    # than ranks (ranks share devices, the barrier / max-over-ranks reduction goes through gloo) -- not a measurement
    backend = os.environ.get("WMD_BENCH_BACKEND", "nccl")
    if os.environ.get("WMD_BENCH_SHARE_DEVICES", "0") == "1":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from wavelet_monodepth_amd import _lib, tuner
    _lib.lib()  # fail loudly if the HIP library is missing
    # tile/split-K choices measured for this workload in an earlier run (any missing key is tuned in the warm-up)
    # (WMD_BENCH_RETUNE=1 ignores the committed choices: used to regenerate that file after kernel changes)
    if os.environ.get("WMD_BENCH_RETUNE", "0") != "1":
        tuner.preload(os.path.join(ROOT, "profiles", "r01_tune_cache_config2.json"))

    if args.workload == "train":
        train_main(args, rank, local_rank, world, dev)
        if world > 1:
            dist.destroy_process_group()
        return

    dec, feats = build_model(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            dec(feats)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step spread (extra fields only)
        barrier()
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(args.steps):
            out = dec(feats)
            marks[k + 1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    if world > 1:
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert all(torch.isfinite(v).all() for v in out.values())

    # kernel-level roofline: the same K steps again with hipEvent pairs around every launch
    roof = None
    if rank == 0:
        with torch.no_grad():
            dec.enable_graph(False)   # per-launch hipEvents need eager launches
            dec(feats)
            _lib.profile_begin()
            for _ in range(args.steps):
                dec(feats)
            recs = _lib.profile_end()
        # trunk convolutions: the direct kernels and the Winograd F(2x2,3x3) ones (the fused-head GEMM chain is listed apart)
        is_wino = lambda r: r["kernel"].startswith("conv_wino_kernel")
        convs = [r for r in recs if (r["kernel"].startswith("conv_fwd_kernel<") and "fused" not in r["kernel"]) or is_wino(r)]
        dom = max(convs, key=lambda r: r["ms"])
        tot_ms = sum(r["ms"] for r in recs)
        conv_ms = sum(r["ms"] for r in convs)
        conv_fl = sum(r["flops"] for r in convs)
        # `flops` are ALGORITHMIC (2*Cin*9*Cout per output pixel); a Winograd kernel executes 2.25x fewer on the matrix pipe
        executed = lambda r: r["flops"] / (2.25 if is_wino(r) else 1.0)
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        ach_exec = executed(dom) / (dom["ms"] * 1e-3) / 1e12
        traffic = None   # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), same kernel only
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f)["kernels"].get(dom["kernel"], {}).get("traffic_bytes_per_launch")
        except OSError:
            pass
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F32_MFMA, 4), "traffic": traffic,
                "algorithm": "winograd F(2x2,3x3): achieved counts algorithmic (direct-convolution) FLOPs, the matrix pipe "
                             "executes 1/2.25 of them" if is_wino(dom) else "direct implicit GEMM",
                "executed_mfma_tflops": round(ach_exec, 2), "frac_executed": round(ach_exec / PEAK_F32_MFMA, 4),
                "algorithmic_bytes_per_launch": dom["bytes"] / dom["calls"],
                "kernel": dom["kernel"], "launches_per_step": dom["calls"] // args.steps,
                "avg_launch_us": round(dom["ms"] * 1e3 / dom["calls"], 2),
                "flop_per_launch": dom["flops"] / dom["calls"],
                "all_conv_kernels": {"achieved": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                                     "frac": round(conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA, 4),
                                     "executed_mfma_tflops": round(sum(executed(r) for r in convs) / (conv_ms * 1e-3) / 1e12, 2),
                                     "share_of_gpu_time": round(conv_ms / tot_ms, 3)},
                "whole_step": {"achieved": round(FLOP_PER_FRAME * BATCH * args.steps / (tot_ms * 1e-3) / 1e12, 2),
                               "gpu_ms_per_step": round(tot_ms / args.steps, 4)},
                "kernels_ms_per_step": {r["kernel"]: round(r["ms"] / args.steps, 4) for r in recs}}

    if rank == 0:
        frames = BATCH * args.steps * world
        res = {
            "metric": "decoder+IDWT frames/sec @640x192 bs12",
            "value": round(frames / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_p10_median_p90": [round(step_ms[int(q * (len(step_ms) - 1))], 4) for q in (0.1, 0.5, 0.9)],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "KITTI ResNet18 640x192 dense wavelet decoder + 4x Haar IDWT, forward, batch 12 per GPU "
                                   "(BASELINE.json configs[1]); encoder features resident in HBM",
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "parallelism": "dp%d (independent shards)" % world},
            "roofline": roof,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1) else cpu_baseline(dec, feats),   # rank 0, N=1 only
        }
        print(json.dumps(res))
    if world > 1:
        barrier()   # rank 0's per-launch roofline pass ends before any rank tears the communicator down
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
