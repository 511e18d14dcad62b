"""CPU, world_size 2 over gloo: the bucketed gradient exchange (wavelet_monodepth_amd/ddp.py).
Sum of per-rank gradients / world == gradient of the concatenated batch; bucket order is decoder-first;
unused parameters (grad never produced) do not stall the exchange."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyNet(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.encoder = nn.Sequential()
        self.encoder.layer1 = nn.Conv2d(3, 8, 3, padding=1)
        self.encoder.layer2 = nn.Conv2d(8, 8, 3, padding=1)
        self.encoder.fc = nn.Linear(8, 4)          # never used: its gradient stays None, like resnet.fc
        self.decoder = nn.Conv2d(8, 1, 3, padding=1)

    def forward(self, x):
        return self.decoder(torch.relu(self.encoder.layer2(torch.relu(self.encoder.layer1(x)))))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wavelet_monodepth_amd.ddp import GradientExchange, bucket_groups
    net = TinyNet()
    if rank == 1:                                        # replicas start different: construction broadcasts rank 0's values
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    from wavelet_monodepth_amd.ddp import _GroupStore
    st = _GroupStore()                                   # how the RCCL backend's unique id travels when a process group is up
    if rank == 0:
        st.set("wmd_comm_uid", b"\x01uid-of-rank-0\x00")
    assert st.get("wmd_comm_uid") == b"\x01uid-of-rank-0\x00"
    gx = GradientExchange(bucket_groups(net.encoder, net.decoder, bucket_bytes=1200), backend="torch", modules=[net])
    assert [b["name"] for b in gx.buckets][0] == "decoder" and len(gx.buckets) >= 3
    ref = TinyNet()
    assert all(torch.equal(a, b) for a, b in zip(net.parameters(), ref.parameters()))
    torch.manual_seed(1)
    full = torch.randn(4, 3, 8, 8)
    shard = full[rank * 2:(rank + 1) * 2]
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    for step in range(3):                                # later steps exercise bucket re-arming with every zero_grad flavour
        if step == 1:
            opt.zero_grad(set_to_none=False)             # keeps the bucket views (used to skip the exchange silently)
        else:
            gx.zero_grad()
        loss = net(shard).pow(2).mean()
        loss.backward()
        gx.finish()
        if step == 0:                                    # the bucket holding the unused `fc` learned not to wait for it
            assert all(b["arm"] <= len(b["params"]) for b in gx.buckets)
            assert sum(b["arm"] for b in gx.buckets) == sum(len(b["params"]) for b in gx.buckets) - 2
    # gradient accumulation: two local backward passes, one exchange
    gx.zero_grad()
    with gx.no_sync():
        net(shard[:1]).pow(2).mean().mul(0.5).backward()
    net(shard[1:]).pow(2).mean().mul(0.5).backward()
    gx.finish()
    grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    # what bench.py's `train` object reports about the exchange itself: the communicator each rank went through and the
    # stand-alone all-reduce time / bus bandwidth of every bucket (scratch buffers: the gradients must be untouched)
    info = gx.comm_info()
    assert info["comm_world"] == world and info["comm_rank"] == rank and info["backend"].startswith("torch")
    timing = gx.bucket_timing(reps=1)
    assert list(timing) == [b["name"] for b in gx.buckets]
    assert all(v["bytes"] == 4 * b["flat"].numel() and v["ms"] > 0 for v, b in zip(timing.values(), gx.buckets))
    assert all(torch.equal(grads[n], p.grad) for n, p in net.named_parameters() if p.grad is not None)
    torch.save(grads, os.path.join(out_dir, "g%d.pt" % rank))
    gx.close()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_full_batch_gradient(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(os.path.join(tmp_path, "g0.pt"))
    g1 = torch.load(os.path.join(tmp_path, "g1.pt"))
    net = TinyNet()
    torch.manual_seed(1)
    full = torch.randn(4, 3, 8, 8)
    # mean over the full batch == mean of the two shard means (equal shard sizes)
    net(full).pow(2).mean().backward()
    for n, p in net.named_parameters():
        if p.grad is None:
            assert float(g0[n].abs().max()) == 0.0       # unused parameter: zero gradient, exchange still completed
            continue
        assert torch.allclose(g0[n], g1[n], atol=0, rtol=0), n       # identical on every rank
        assert torch.allclose(g0[n], p.grad, atol=1e-6, rtol=1e-5), n


def _mismatch_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wavelet_monodepth_amd.ddp import GradientExchange, bucket_groups
    net = TinyNet()
    x = torch.randn(2, 3, 8, 8)

    def loss_of(net, use_fc):       # rank 1 reaches `encoder.fc`, rank 0 does not: a data-dependent graph
        y = net(x).pow(2).mean()
        return y + net.encoder.fc(torch.ones(1, 8)).sum() if use_fc else y
    res = []
    for static in (True, False):
        gx = GradientExchange(bucket_groups(net.encoder, net.decoder, bucket_bytes=1200), backend="torch", modules=[net],
                              static_graph=static)
        gx.zero_grad()
        loss_of(net, rank == 1).backward()
        try:
            gx.finish()
            res.append("ok")
        except RuntimeError as e:
            res.append("raised" if "different parameter sets" in str(e) else "other: %s" % e)
        arms = [b["arm"] for b in gx.buckets]
        full = [len(b["params"]) for b in gx.buckets]
        res.append(arms == full)
        gx.close()
        for p in net.parameters():
            p.grad = None
    torch.save(res, os.path.join(out_dir, "m%d.pt" % rank))
    dist.destroy_process_group()


def test_ranks_that_reach_different_parameters_are_caught_before_any_arm_is_lowered(tmp_path):
    """Round-2 ADVICE: a bucket's arm used to be learned from the LOCAL arrival count; ranks with different used-parameter
    sets then issued their all-reduces in different orders.  static_graph=True: the first finish() compares the counts over
    the ranks and raises on every rank, arms untouched; static_graph=False: no arm is ever lowered and the step completes."""
    world, port = 2, _free_port()
    mp.spawn(_mismatch_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, "m%d.pt" % r))
        assert got == ["raised", True, "ok", True], (r, got)


def _strong_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wavelet_monodepth_amd.ddp import GradientExchange, bucket_groups
    net = TinyNet()
    gx = GradientExchange(bucket_groups(net.encoder, net.decoder, bucket_bytes=1200), backend="torch", modules=[net])
    torch.manual_seed(7)
    full, target = torch.randn(4, 3, 8, 8), torch.randn(4, 1, 8, 8)
    per = full.shape[0] // world                          # `bench.py --workload train --strong`: the GLOBAL batch split over the ranks
    x, y = full[rank * per:(rank + 1) * per], target[rank * per:(rank + 1) * per]
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    losses = []
    for _ in range(3):
        gx.zero_grad()
        loss = (net(x) - y).abs().mean()
        loss.backward()
        gx.finish()                                       # sum over the ranks / world = the full batch's mean gradient
        opt.step()
        t = loss.detach().clone()
        dist.all_reduce(t)
        losses.append(float(t) / world)                   # equal shards: the global loss is the mean of the shard losses
    torch.save({"losses": losses, "params": [p.detach().clone() for p in net.parameters()]}, os.path.join(out_dir, "s%d.pt" % rank))
    gx.close()
    dist.destroy_process_group()


def test_strong_scaling_split_reproduces_the_single_rank_trajectory(tmp_path):
    """a fixed global batch split over 2 ranks (what `bench.py --strong` does) must walk the SAME loss trajectory
    and reach the same parameters as one rank stepping the whole batch -- three optimizer steps with momentum, so that an
    exchange that averaged wrongly (or a bucket that stopped being sent after the first step) shows up in steps 2 and 3."""
    world, port = 2, _free_port()
    mp.spawn(_strong_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    net = TinyNet()
    torch.manual_seed(7)
    full, target = torch.randn(4, 3, 8, 8), torch.randn(4, 1, 8, 8)
    opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9)
    want = []
    for _ in range(3):
        opt.zero_grad()
        loss = (net(full) - target).abs().mean()
        loss.backward()
        opt.step()
        want.append(float(loss))
    for r in range(world):
        got = torch.load(os.path.join(tmp_path, "s%d.pt" % r))
        assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(got["losses"], want)), (r, got["losses"], want)
        for a, b in zip(got["params"], net.parameters()):
            assert torch.allclose(a, b.detach(), atol=1e-6, rtol=1e-5)


def test_message_sizes_match_survey():
    """R18 encoder + wavelet decoder: 11.2 M + 3.36 M parameters -> ~58 MB of gradients."""
    import numpy as np
    from wavelet_monodepth_amd.ddp import GradientExchange, monodepth_groups
    from wavelet_monodepth_amd.encoders import ResnetEncoder
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    enc = ResnetEncoder(18)
    dec = DepthWaveProgressiveDecoder(enc.num_ch_enc)
    gx = GradientExchange(monodepth_groups(enc, dec), world=1, rank=0, backend="torch")
    sizes = gx.message_bytes()
    assert list(sizes)[0] == "decoder" and sizes["decoder"] == 4 * 3361625
    assert 55e6 < sum(sizes.values()) < 62e6
    gx.close()
