"""GPU parity of the hand-written backward kernels (dgrad / wgrad / bias / head / IDWT adjoint) against
autograd of the CPU oracle, and of a whole decoder training step against the reference's gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import decoder_ref as R
from wavelet_monodepth_amd import synth
from util import R18, assert_close, key_str, kitti_feats, load_golden, sample, t

pytestmark = pytest.mark.gpu
GRAD_TOL = 5e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


BWD_CASES = [
    # B, C1, C2, up, Cout, H, W, k, pad, act
    (2, 5, 0, 1, 7, 6, 10, 3, "reflect", "none"),
    (2, 19, 0, 1, 7, 5, 8, 3, "zero", "elu"),
    (2, 16, 8, 2, 19, 4, 6, 3, "reflect", "elu"),
    (1, 32, 64, 2, 32, 12, 40, 3, "reflect", "elu"),
    (2, 24, 0, 1, 40, 6, 20, 3, "replicate", "leaky"),
    (1, 23, 10, 2, 13, 30, 40, 3, "reflect", "leaky"),
    (3, 8, 0, 1, 16, 2, 2, 3, "reflect", "elu"),
    (2, 21, 0, 1, 9, 5, 7, 1, "zero", "leaky"),
    (1, 64, 0, 1, 64, 24, 80, 1, "zero", "leaky"),
    (1, 64, 64, 2, 64, 48, 160, 3, "reflect", "elu"),
]


@pytest.mark.parametrize("case", BWD_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv_backward(dev, case):
    from wavelet_monodepth_amd import ops
    B, C1, C2, up, Cout, H, W, k, pad, act = case
    x1 = t(synth.normal((B, C1, H // up, W // up), "bx1", 3)).requires_grad_(True)
    x2 = t(synth.normal((B, C2, H, W), "bx2", 3)).requires_grad_(True) if C2 else None
    w, b = [t(a).requires_grad_(True) for a in synth.conv_params("bw", Cout, C1 + C2, k, 3)]
    gy = t(synth.normal((B, Cout, H, W), "bgy", 3))
    xin = R.up2(x1) if up == 2 else x1
    if x2 is not None:
        xin = torch.cat([xin, x2], 1)
    ref = R.conv3x3(xin, w, b, pad) if k == 3 else R.conv1x1(xin, w, b)
    slope = 0.1
    ref = {"none": lambda v: v, "elu": torch.nn.functional.elu, "leaky": lambda v: torch.nn.functional.leaky_relu(v, slope)}[act](ref)
    (ref * gy).sum().backward()

    gx1 = x1.detach().to(dev).requires_grad_(True)
    gx2 = x2.detach().to(dev).requires_grad_(True) if C2 else None
    gw, gb = w.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    y = ops.conv2d_fused(gx1, gw, gb, x2=gx2, up1=up, pad=pad, act=act, slope=slope)
    (y * gy.to(dev)).sum().backward()
    assert_close(gx1.grad, x1.grad, GRAD_TOL, "dx1")
    if C2:
        assert_close(gx2.grad, x2.grad, GRAD_TOL, "dx2")
    assert_close(gw.grad, w.grad, GRAD_TOL, "dw")
    assert_close(gb.grad, b.grad, GRAD_TOL, "db")


@pytest.mark.parametrize("mode,cout,pad", [(0, 3, "zero"), (1, 1, "reflect"), (2, 3, "reflect")])
def test_head3x3_backward(dev, mode, cout, pad):
    from wavelet_monodepth_amd import ops
    B, C, H, W = 2, 20, 12, 40
    xp = t(synth.normal((B, C, H, W), "hbxp", 4)).requires_grad_(True)
    xn = t(synth.normal((B, C, H, W), "hbxn", 4)).requires_grad_(True)
    wp, bp = [t(a).requires_grad_(True) for a in synth.conv_params("hbwp", cout, C, 3, 4)]
    wn, bn = [t(a).requires_grad_(True) for a in synth.conv_params("hbwn", cout, C, 3, 4)]
    gy = t(synth.normal((B, cout, H, W), "hbgy", 4))
    scale = 2.0
    cp = R.conv3x3(xp, wp, bp, pad)
    if mode == 0:
        ref = scale * cp
    elif mode == 1:
        ref = scale * torch.sigmoid(cp)
    else:
        ref = scale * torch.sigmoid(cp) - scale * torch.sigmoid(R.conv3x3(xn, wn, bn, pad))
    (ref * gy).sum().backward()
    d = lambda v: v.detach().to(dev).requires_grad_(True)
    gxp, gxn, gwp, gbp, gwn, gbn = d(xp), d(xn), d(wp), d(bp), d(wn), d(bn)
    y = ops.head3x3(gxp, gwp, gbp, gxn if mode == 2 else None, gwn if mode == 2 else None, gbn if mode == 2 else None,
                    pad=pad, mode=mode, scale=scale)
    (y * gy.to(dev)).sum().backward()
    assert_close(gxp.grad, xp.grad, GRAD_TOL, "dxp")
    assert_close(gwp.grad, wp.grad, GRAD_TOL, "dwp")
    assert_close(gbp.grad, bp.grad, GRAD_TOL, "dbp")
    if mode == 2:
        assert_close(gxn.grad, xn.grad, GRAD_TOL, "dxn")
        assert_close(gwn.grad, wn.grad, GRAD_TOL, "dwn")
        assert_close(gbn.grad, bn.grad, GRAD_TOL, "dbn")


def test_kitti_dense_decoder_gradients_vs_reference_golden(dev):
    """loss = sum_s mean(disp_s): gradients w.r.t. all five feature maps and every parameter, against
    the gradients the REFERENCE module produced for the same synth weights/inputs."""
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    g = load_golden("kitti_dense_r18_64x64_grads.npz")
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev)
    feats = [f.to(dev).requires_grad_(True) for f in kitti_feats(2, 64, 64)]
    out = dec(feats)
    loss = sum(out[("disp", s)].mean() for s in range(4))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    for k, f in enumerate(feats):
        assert_close(f.grad, g["dfeat%d" % k], 1e-4, "dfeat%d" % k)
    n = 0
    for name, p in dec.named_parameters():
        assert p.grad is not None, name
        assert_close(sample(p.grad.cpu().numpy()), g["d|" + name], 1e-4, name)
        n += 1
    assert n == 52  # 8 ConvBlocks x (w, b) + 9 heads x (w1, b1, w3, b3)


def test_training_steps_track_the_oracle(dev):
    """Four Adam steps (KITTI trainer's param groups over `.convs`, trainer.py:74-75,96-98) through the HIP
    forward+backward follow the same loss trajectory as the CPU oracle under torch.autograd."""
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=2).to(dev)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and "inverse_wt" not in k)
          for k, v in dec.state_dict().items()}
    groups = []
    for key, m in dec.convs.items():
        groups.append({"params": [p for n, p in m.named_parameters() if n.endswith("weight")], "weight_decay": 1e-5})
        groups.append({"params": [p for n, p in m.named_parameters() if n.endswith("bias")], "weight_decay": 0.0})
    opt = torch.optim.Adam(groups, lr=1e-4)
    names = [n for n, _ in dec.named_parameters()]
    opt_ref = torch.optim.Adam([{"params": [sd[n] for n in names if n.endswith("weight")], "weight_decay": 1e-5},
                                {"params": [sd[n] for n in names if n.endswith("bias")], "weight_decay": 0.0}], lr=1e-4)
    feats = kitti_feats(2, 64, 96, seed=2)
    gfeats = [f.to(dev) for f in feats]
    for step in range(4):
        out = dec(gfeats)
        loss = sum(((out[("disp", s)] - 0.25) ** 2).mean() for s in range(4))
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref = R.kitti_wave_decoder(feats, sd)
        loss_ref = sum(((ref[("disp", s)] - 0.25) ** 2).mean() for s in range(4))
        opt_ref.zero_grad()
        loss_ref.backward()
        opt_ref.step()
        assert abs(float(loss) - float(loss_ref)) < 2e-5 * max(1.0, abs(float(loss_ref))), (step, float(loss), float(loss_ref))
    # Adam divides by sqrt(v): elements with near-zero gradients amplify rounding noise into O(lr) differences,
    # so parameters are compared on the scale of the total movement (4 * lr = 4e-4 absolute)
    for n, p in dec.named_parameters():
        assert float((p.detach().cpu() - sd[n].detach()).abs().max()) < 4e-5, "parameter after 4 steps: " + n


def test_rccl_exchange_single_rank_and_encoder_decoder_step(dev):
    """wmd_comm_* (RCCL) on a world of one + a full encoder/decoder training step with bucketed gradients:
    the exchange must be the identity and leave the same gradients as plain autograd."""
    from wavelet_monodepth_amd.ddp import GradientExchange, monodepth_groups
    from wavelet_monodepth_amd.encoders import ResnetEncoder
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder

    class LocalStore(dict):
        def set(self, k, v):
            self[k] = v

        def get(self, k):
            return self[k]

    torch.manual_seed(0)
    enc = ResnetEncoder(18).to(dev)
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(enc.num_ch_enc), seed=1).to(dev)
    img = t(synth.uniform((2, 3, 64, 96), "img", 0, 0.0, 1.0)).to(dev)

    def loss_fn():
        out = dec(enc(img))
        return sum(out[("disp", s)].mean() for s in range(4))

    loss_fn().backward()
    ref = {n: p.grad.clone() for n, p in list(enc.named_parameters()) + list(dec.named_parameters()) if p.grad is not None}
    for p in list(enc.parameters()) + list(dec.parameters()):
        p.grad = None
    gx = GradientExchange(monodepth_groups(enc, dec), world=1, rank=0, backend="rccl", store=LocalStore())
    gx.zero_grad()
    loss_fn().backward()
    gx.finish()
    torch.cuda.synchronize()
    for n, p in list(enc.named_parameters()) + list(dec.named_parameters()):
        if n in ref:
            assert_close(p.grad, ref[n], 2e-5, n)  # MIOpen encoder kernels are not bitwise run-to-run stable
    assert [b["name"] for b in gx.buckets][0] == "decoder"
    # second step with the optimizer's own zero_grad(set_to_none=False): buckets were re-armed by finish()
    torch.optim.SGD(list(enc.parameters()) + list(dec.parameters()), lr=0.0).zero_grad(set_to_none=False)
    loss_fn().backward()
    gx.finish()
    torch.cuda.synchronize()
    for n, p in list(enc.named_parameters()) + list(dec.named_parameters()):
        if n in ref:
            assert_close(p.grad, ref[n], 2e-5, "step 2 " + n)
    # wmd_comm_broadcast (world of one: the identity) through the packing path used at construction
    before = [p.detach().clone() for p in dec.parameters()]
    gx._broadcast_tensors(list(dec.parameters()))
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, dec.parameters()))
    gx.close()


def _rccl_worker(rank, world, port, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from wavelet_monodepth_amd.ddp import GradientExchange, bucket_groups
    from wavelet_monodepth_amd.encoders import ResnetEncoder
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    torch.manual_seed(100 + rank)                               # replicas start DIFFERENT: construction must broadcast rank 0's
    enc = ResnetEncoder(18).to(dev).eval()                      # eval: BatchNorm uses running stats -> shards are independent
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(enc.num_ch_enc), seed=1 + rank).to(dev)
    store = dist.TCPStore("127.0.0.1", port, world, rank == 0)
    gx = GradientExchange(bucket_groups(enc, dec, bucket_bytes=8 << 20), world=world, rank=rank, backend="rccl", store=store,
                          modules=[enc, dec])
    full = t(synth.uniform((2 * world, 3, 64, 96), "ddp_img", 0, 0.0, 1.0))
    shard = full[2 * rank:2 * rank + 2].to(dev)
    for _ in range(2):
        gx.zero_grad()
        out = dec(enc(shard))
        sum((out[("disp", s)] ** 2).mean() for s in range(4)).backward()
        gx.finish()
    torch.cuda.synchronize()
    state = {"grads": {n: p.grad.cpu() for n, p in list(enc.named_parameters()) + list(dec.named_parameters()) if p.grad is not None},
             "params": {n: p.detach().cpu() for n, p in list(enc.named_parameters()) + list(dec.named_parameters())}}
    torch.save(state, os.path.join(out_dir, "r%d.pt" % rank))
    gx.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the 8-GPU scaling node); one-GPU boxes skip")
def test_rccl_two_rank_exchange_equals_full_batch_gradient(dev, tmp_path):
    """Two processes, one GPU each, RCCL over xGMI: rank 0's parameters reach rank 1 at construction, and the averaged shard
    gradients equal the gradient of the concatenated batch computed by one process."""
    import socket
    import torch.multiprocessing as mp
    from wavelet_monodepth_amd.encoders import ResnetEncoder
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.start_processes(_rccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in (0, 1))
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), "replicas differ after the construction broadcast: " + n
    for n in r0["grads"]:
        assert torch.equal(r0["grads"][n], r1["grads"][n]), "all-reduced gradients differ between ranks: " + n
    torch.manual_seed(100)
    enc = ResnetEncoder(18).to(dev).eval()
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(enc.num_ch_enc), seed=1).to(dev)
    full = t(synth.uniform((4, 3, 64, 96), "ddp_img", 0, 0.0, 1.0)).to(dev)
    out = dec(enc(full))
    sum((out[("disp", s)] ** 2).mean() for s in range(4)).backward()
    for n, p in list(enc.named_parameters()) + list(dec.named_parameters()):
        if p.grad is not None:
            assert_close(r0["grads"][n], p.grad, 5e-5, n)
