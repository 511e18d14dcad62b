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


# ---- round 2: Winograd weight gradient, activation-derivative gating, stacked heads ---------------------------------------
WINO_WGRAD_CASES = [
    # B, C1, C2, up, Cout, H, W, pad
    (2, 19, 0, 1, 7, 6, 10, "zero"),            # ragged channels, map smaller than a tile row
    (1, 32, 64, 2, 32, 12, 40, "reflect"),      # upconv(1,1) structure (fused upsample + concat)
    (2, 24, 0, 1, 40, 7, 21, "replicate"),      # odd sizes: tile overhang in both directions
    (1, 64, 0, 1, 3, 16, 48, "reflect"),        # a head's Cout = 3 filter (16-row out-channel tiles)
    (3, 8, 0, 1, 16, 2, 2, "reflect"),          # smallest legal reflect size
]


@pytest.mark.parametrize("case", WINO_WGRAD_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_wgrad_winograd_every_configuration_vs_oracle(dev, case):
    """conv_wgrad_wino_kernel: every entry of the Winograd F(2x2,3x3) weight-gradient table forced through
    wmd_conv_wgrad_args.tune_cfg (and two pixel splits), plus the direct kernel (tune_cfg = -1), against autograd through
    the oracle's convolution.  dU = sum_tiles (A dY A^T) (x) (B^T d B), dg = G^T dU G is exact algebra: 2e-5 like the direct
    form."""
    import ctypes as C
    from wavelet_monodepth_amd import _lib
    B, C1, C2, up, Cout, H, W, pad = case
    x1 = t(synth.normal((B, C1, H // up, W // up), "wwx1", 5)).requires_grad_(False)
    x2 = t(synth.normal((B, C2, H, W), "wwx2", 5)) if C2 else None
    w = t(synth.normal((Cout, C1 + C2, 3, 3), "www", 5)).requires_grad_(True)
    b = t(synth.normal((Cout,), "wwb", 5)).requires_grad_(True)
    dz = t(synth.normal((B, Cout, H, W), "wwdz", 5))
    xin = R.up2(x1) if up == 2 else x1
    if x2 is not None:
        xin = torch.cat([xin, x2], 1)
    (R.conv3x3(xin, w, b, pad) * dz).sum().backward()
    l = _lib.lib()
    g = lambda v: None if v is None else v.to(dev)
    x1d, x2d, dzd = g(x1), g(x2), g(dz)
    names = [l.wmd_conv_wgrad_config_name(i).decode() for i in range(l.wmd_conv_wgrad_num_configs())]
    assert len(names) >= 9 and all(n.startswith("conv_wgrad_wino") for n in names)
    tested = 0
    for cfg in [-1] + list(range(1, len(names) + 1)):
        for ns in (0, 3):
            dw = torch.full((Cout, C1 + C2, 3, 3), float("nan"), device=dev)
            db = torch.full((Cout,), float("nan"), device=dev)
            a = _lib.ConvWgradArgs(B=B, H=H, W=W, C1=C1, up1=up, C2=C2, Cout=Cout, ksize=3, pad_mode=_lib.PAD[pad], x1=x1d.data_ptr(),
                                   x2=None if x2d is None else x2d.data_ptr(), dz=dzd.data_ptr(), dw=dw.data_ptr(), dbias=db.data_ptr(),
                                   workspace=None, workspace_floats=0, tune_cfg=cfg, tune_nsplit=ns)
            n = l.wmd_conv_wgrad_workspace_floats(C.byref(a))
            ws = torch.empty(max(n, 1), device=dev)
            a.workspace, a.workspace_floats = ws.data_ptr(), n
            _lib.check(l.wmd_conv_wgrad(C.byref(a), torch.cuda.current_stream().cuda_stream), "cfg %d" % cfg)
            what = "direct" if cfg < 0 else names[cfg - 1]
            assert_close(dw, w.grad, 2e-5, "dW %s nsplit %d" % (what, ns))
            assert_close(db, b.grad, 2e-5, "db %s nsplit %d" % (what, ns))
            tested += 1
    assert tested == 2 * (len(names) + 1)


def test_activation_gating_equals_separate_activation_backward(dev):
    """conv -> ELU -> {conv 3x3 (upsampled + skip), conv 1x1}: with x1_gate on both consumers and grad_is_dz on the producer
    (what the decoders do in training) every gradient equals the ungated graph's and the oracle's."""
    from wavelet_monodepth_amd import ops
    x = t(synth.normal((2, 12, 6, 10), "gx", 8)).requires_grad_(True)
    skip = t(synth.normal((2, 5, 12, 20), "gs", 8)).requires_grad_(True)
    (w0, b0), (w1, b1), (w2, b2) = [[t(a).requires_grad_(True) for a in synth.conv_params(n, co, ci, k, 8)]
                                    for n, co, ci, k in (("g0", 16, 12, 3), ("g1", 7, 21, 3), ("g2", 9, 16, 1))]
    elu = torch.nn.functional.elu
    y = elu(R.conv3x3(x, w0, b0, "reflect"))
    o1 = torch.nn.functional.leaky_relu(R.conv3x3(torch.cat([R.up2(y), skip], 1), w1, b1, "reflect"), 0.1)
    o2 = R.conv1x1(y, w2, b2)
    ((o1 ** 2).sum() + (o2 ** 2).sum()).backward()
    leaves = [x, skip, w0, b0, w1, b1, w2, b2]
    for gated in (False, True):
        d = [v.detach().to(dev).requires_grad_(True) for v in leaves]
        gx, gs, gw0, gb0, gw1, gb1, gw2, gb2 = d
        gate = ("elu", 0.0) if gated else None
        yy = ops.conv2d_fused(gx, gw0, gb0, pad="reflect", act="elu", grad_is_dz=gated)
        p1 = ops.conv2d_fused(yy, gw1, gb1, x2=gs, up1=2, pad="reflect", act="leaky", slope=0.1, x1_gate=gate)
        p2 = ops.conv2d_fused(yy, gw2, gb2, pad="zero", x1_gate=gate)
        ((p1 ** 2).sum() + (p2 ** 2).sum()).backward()
        for a_, b_, name in zip(d, leaves, ("x", "skip", "w0", "b0", "w1", "b1", "w2", "b2")):
            assert_close(a_.grad, b_.grad, GRAD_TOL, "%s (gated=%s)" % (name, gated))


def test_stacked_heads_equal_per_head_operators_and_the_oracle(dev):
    """ops.stacked_heads (one launch per stage over the +, - and LL heads) against the per-head operator path and against
    autograd through the oracle, values and every gradient."""
    from wavelet_monodepth_amd import ops
    C, H, W, B = 64, 10, 24, 2
    x = t(synth.normal((B, C, H, W), "shx", 9))
    mk = lambda tag, mid, out: [t(a) for a in synth.conv_params(tag + "1", mid, C, 1, 9)] + [t(a) for a in synth.conv_params(tag + "3", out, mid, 3, 9)]
    hp, hn, hl = mk("shp", C, 3), mk("shn", C, 3), mk("shl", C // 4, 1)
    gyh = t(synth.normal((B, 3, H, W), "shgy", 9))
    gyl = t(synth.normal((B, 1, H, W), "shgl", 9))
    lk = lambda v: torch.nn.functional.leaky_relu(v, 0.1)

    def oracle(xx, p, n, ll):
        sig = lambda h: torch.sigmoid(R.conv3x3(lk(R.conv1x1(xx, h[0], h[1])), h[2], h[3], "reflect"))
        return 4.0 * (sig(p) - sig(n)), 16.0 * sig(ll)

    leaves = [x] + hp + hn + hl
    ref = [v.clone().requires_grad_(True) for v in leaves]
    yh_r, yl_r = oracle(ref[0], ref[1:5], ref[5:9], ref[9:13])
    ((yh_r * gyh).sum() + (yl_r * gyl).sum()).backward()
    d = [v.to(dev).requires_grad_(True) for v in leaves]
    yh, yl = ops.stacked_heads(d[0], d[1:5], d[5:9], 4.0, head_ll=d[9:13], scale_ll=16.0)
    assert float((yh.cpu() - yh_r).abs().max()) < 2e-5 and float((yl.cpu() - yl_r).abs().max()) < 4e-5
    ((yh * gyh.to(dev)).sum() + (yl * gyl.to(dev)).sum()).backward()
    for a_, b_, k in zip(d, ref, range(13)):
        assert_close(a_.grad, b_.grad, GRAD_TOL, "stacked heads: leaf %d" % k)
    # two heads only (levels 3..1)
    d2 = [v.to(dev).requires_grad_(True) for v in leaves[:9]]
    yh2, none = ops.stacked_heads(d2[0], d2[1:5], d2[5:9], 4.0)
    assert none is None and float((yh2 - yh).abs().max()) < 1e-6


@pytest.mark.parametrize("C,H,W,B,with_ll", [(32, 12, 40, 2, False), (32, 7, 9, 1, False), (64, 6, 10, 2, False), (128, 4, 6, 2, False),
                                             (256, 6, 20, 2, True), (256, 2, 2, 1, True)])
def test_fused_training_level_vs_oracle_and_stacked_path(dev, C, H, W, B, with_ll):
    """ops.fused_level_train (round 5: a level's heads + Haar synthesis in training mode on the inference kernels, with the
    1x1 outputs and the sigmoid outputs written for the hand-written backward) against autograd through the oracle -- values,
    the `mid` tensor, and the gradient of every input and parameter under a loss that reaches all four outputs (yh, the
    low-pass head's own output, the synthesis and the clamped disparity) -- and against the stacked-operator path it replaces."""
    from wavelet_monodepth_amd import ops
    assert ops.fused_train_supported(C, H, W, with_ll)
    x = t(synth.normal((B, C, H, W), "ftx", 5))
    mk = lambda tag, mid, out: [t(a) for a in synth.conv_params(tag + "1", mid, C, 1, 5)] + [t(a) for a in synth.conv_params(tag + "3", out, mid, 3, 5)]
    hp, hn = mk("ftp", C, 3), mk("ftn", C, 3)
    hl = mk("ftl", C // 4, 1) if with_ll else []
    yl0 = t(synth.uniform((B, 1, H, W), "ftyl", 5, 2.0, 9.0))
    s_hf, s_ll, dsc = 4.0, 16.0, 0.3
    g = {k: t(synth.normal(shape, "ftg" + k, 5)) for k, shape in (("yh", (B, 3, H, W)), ("ll", (B, 1, H, W)), ("out", (B, 1, 2 * H, 2 * W)),
                                                                   ("disp", (B, 1, 2 * H, 2 * W)))}
    lk = lambda v: torch.nn.functional.leaky_relu(v, 0.1)

    def oracle(xx, p, n, ll, yl):
        sig = lambda h: torch.sigmoid(R.conv3x3(lk(R.conv1x1(xx, h[0], h[1])), h[2], h[3], "reflect"))
        yh = s_hf * (sig(p) - sig(n))
        lo = s_ll * sig(ll) if ll else yl
        out = R.haar_idwt(lo, yh.unsqueeze(1))
        return yh, lo, out, torch.clamp(out * dsc, 0, 1)

    leaves = [x] + hp + hn + hl + ([] if with_ll else [yl0])
    ref = [v.clone().requires_grad_(True) for v in leaves]
    r_yh, r_lo, r_out, r_disp = oracle(ref[0], ref[1:5], ref[5:9], ref[9:13] if with_ll else None, None if with_ll else ref[9])
    loss_r = (r_yh * g["yh"]).sum() + (r_out * g["out"]).sum() + (r_disp * g["disp"]).sum() + ((r_lo * g["ll"]).sum() if with_ll else 0.0)
    loss_r.backward()
    assert float(((r_out * dsc <= 0) | (r_out * dsc >= 1)).float().mean()) > 0.01, "the clamp must be active somewhere"

    def run(fn):
        d = [v.to(dev).requires_grad_(True) for v in leaves]
        yh, lo, out, disp = fn(d)
        loss = (yh * g["yh"].to(dev)).sum() + (out * g["out"].to(dev)).sum() + (disp * g["disp"].to(dev)).sum()
        if with_ll:
            loss = loss + (lo * g["ll"].to(dev)).sum()
        loss.backward()
        return d, (yh, lo, out, disp)

    def fused(d):
        yh, lo, out, disp, mid = ops.fused_level_train(d[0], d[1:5], d[5:9], s_hf, yl=None if with_ll else d[9], disp_scale=dsc, clamp01=True,
                                                       head_ll=d[9:13] if with_ll else None, scale_ll=s_ll)
        heads = ([ref[9:13]] if with_ll else []) + [ref[1:5], ref[5:9]]
        want = torch.cat([lk(R.conv1x1(ref[0], h[0], h[1])) for h in heads], 1).detach()
        assert_close(mid, want, 2e-5, "mid (LeakyReLU outputs, order [LL, +, -])")
        return yh, lo, out, disp

    def stacked(d):
        yh, lo = ops.stacked_heads(d[0], d[1:5], d[5:9], s_hf, head_ll=d[9:13] if with_ll else None, scale_ll=s_ll)
        out, disp = ops.idwt_haar(lo if with_ll else d[9], yh.unsqueeze(1), disp_scale=dsc, clamp01=True)
        return yh, lo, out, disp

    d_f, o_f = run(fused)
    for got, want, name in zip(o_f, (r_yh, r_lo, r_out, r_disp), ("yh", "yl", "out", "disp")):
        if got is not None:
            assert_close(got, want.detach(), 2e-5, "fused level: " + name)
    for k, (a_, b_) in enumerate(zip(d_f, ref)):
        assert_close(a_.grad, b_.grad, GRAD_TOL, "fused level: gradient of leaf %d" % k)
    d_s, o_s = run(stacked)
    for k, (a_, b_) in enumerate(zip(d_f, d_s)):
        assert_close(a_.grad, b_.grad, GRAD_TOL, "fused vs stacked path: gradient of leaf %d" % k)


@pytest.mark.parametrize("H,W,B", [(16, 64, 16), (24, 40, 18), (8, 128, 17)])
def test_one_pass_head_backward_at_32_channels_vs_oracle_and_the_launch_set(dev, H, W, B, monkeypatch):
    """Round 6: head_bwd_fused32_kernel (wmd_head_bwd at C = 32: the 3x3 data + weight gradients, the 1x1 data + weight gradients of
    both heads in ONE pass, dz in a wave-private LDS tile) against autograd through the oracle and against the separate launch sets
    (wmd_head3x3_bwd + wmd_head1x1_bwd), on maps where most 16-pixel groups touch an edge (16 x 64), where groups straddle rows
    (24 x 40) and with an ELU gate on x; the kernel must be the one that ran."""
    from wavelet_monodepth_amd import _lib, ops
    C = 32
    x = t(synth.normal((B, C, H, W), "f32x", 5))
    mk = lambda tag, mid, out: [t(a) for a in synth.conv_params(tag + "1", mid, C, 1, 5)] + [t(a) for a in synth.conv_params(tag + "3", out, mid, 3, 5)]
    hp, hn = mk("f32p", C, 3), mk("f32n", C, 3)
    yl0 = t(synth.uniform((B, 1, H, W), "f32yl", 5, 2.0, 9.0))
    g_yh, g_out = t(synth.normal((B, 3, H, W), "f32gyh", 5)), t(synth.normal((B, 1, 2 * H, 2 * W), "f32gout", 5))
    lk = lambda v: torch.nn.functional.leaky_relu(v, 0.1)
    leaves = [x] + hp + hn + [yl0]
    grads, masks = {}, None
    for merged in (True, False):
        monkeypatch.setattr(ops, "_HEAD_BWD_MERGED", merged)
        d = [v.to(dev).requires_grad_(True) for v in leaves]
        xin = torch.nn.functional.elu(d[0]).detach().requires_grad_(True)
        yh, lo, out, disp, mid = ops.fused_level_train(xin, d[1:5], d[5:9], 4.0, yl=d[9], disp_scale=0.3, clamp01=True, x_gate=("elu", 0.0))
        masks = ((mid[:, :C] > 0).cpu(), (mid[:, C:] > 0).cpu())       # the LeakyReLU piece every element took (order [+, -])
        _lib.profile_begin()
        ((yh * g_yh.to(dev)).sum() + (out * g_out.to(dev)).sum()).backward()
        names = {r["kernel"] for r in _lib.profile_end()}
        assert ("head_bwd_fused32_kernel" in names) == merged, names
        grads[merged] = [xin.grad] + [v.grad for v in d[1:]]
    for k, (a_, b_) in enumerate(zip(grads[True], grads[False])):
        assert_close(a_, b_, GRAD_TOL, "one-pass vs launch set: gradient of leaf %d" % k)
    # the oracle differentiates the LeakyReLU piece the device took (the only kink on the path); x is an ELU output whose producer
    # wants dz: the oracle's gradient of the pre-activation leaf is what the kernel's gate returns
    ref = [v.clone().requires_grad_(True) for v in leaves]
    xe = torch.nn.functional.elu(ref[0])

    def sig(h, m):
        z = R.conv1x1(xe, h[0], h[1])
        return torch.sigmoid(R.conv3x3(torch.where(m, z, 0.1 * z), h[2], h[3], "reflect"))

    r_yh = 4.0 * (sig(ref[1:5], masks[0]) - sig(ref[5:9], masks[1]))
    r_out = R.haar_idwt(ref[9], r_yh.unsqueeze(1))
    ((r_yh * g_yh).sum() + (r_out * g_out).sum()).backward()
    for k, (a_, b_) in enumerate(zip(grads[True], [v.grad for v in ref])):
        assert_close(a_, b_, GRAD_TOL, "one-pass head backward vs oracle: gradient of leaf %d" % k)


@pytest.mark.parametrize("C,H,W,B,with_ll", [(32, 9, 21, 2, False), (64, 2, 5, 1, True), (16, 3, 2, 3, False), (80, 7, 66, 1, False), (128, 5, 9, 1, True),
                                             (256, 6, 20, 2, True)])
def test_head3x3_backward_kernels_vs_oracle_and_generic_path(dev, C, H, W, B, with_ll, monkeypatch):
    """wmd_head3x3_bwd (tap-partial rows gathered from dy, one pass over mid for the data and one for the weight gradient,
    reflection ring folded analytically) against autograd through the oracle and against the generic dgrad / wgrad kernels on
    the block-diagonal filter: odd sizes, maps of 2 rows / 2 columns (both ring rows fold onto the same source row), channel
    counts that are not multiples of 16 or 64, more pixels than one wave tile, with and without the low-pass head."""
    from wavelet_monodepth_amd import ops
    x = t(synth.normal((B, C, H, W), "hbx", 5))
    mk = lambda tag, mid, out: [t(a) for a in synth.conv_params(tag + "1", mid, C, 1, 5)] + [t(a) for a in synth.conv_params(tag + "3", out, mid, 3, 5)]
    hp, hn = mk("hbp", C, 3), mk("hbn", C, 3)
    hl = mk("hbl", max(C // 4, 1), 1) if with_ll else []
    gyh, gyl = t(synth.normal((B, 3, H, W), "hbgy", 5)), t(synth.normal((B, 1, H, W), "hbgl", 5))
    lk = lambda v: torch.nn.functional.leaky_relu(v, 0.1)
    sig = lambda xx, h: torch.sigmoid(R.conv3x3(lk(R.conv1x1(xx, h[0], h[1])), h[2], h[3], "reflect"))
    leaves = [x] + hp + hn + hl
    ref = [v.clone().requires_grad_(True) for v in leaves]
    loss = (2.0 * (sig(ref[0], ref[1:5]) - sig(ref[0], ref[5:9])) * gyh).sum()
    if with_ll:
        loss = loss + (8.0 * sig(ref[0], ref[9:13]) * gyl).sum()
    loss.backward()
    got = {}
    monkeypatch.setattr(ops, "_HEAD_BWD_MIN_PIXELS", 0)      # the size rule would send these small maps to the generic kernels
    monkeypatch.setattr(ops, "_HEAD_BWD1_MIN_PIXELS", 0)
    for new_path in (True, False):
        monkeypatch.setattr(ops, "_HEAD_BWD", new_path)
        d = [v.to(dev).requires_grad_(True) for v in leaves]
        yh, yl = ops.stacked_heads(d[0], d[1:5], d[5:9], 2.0, head_ll=d[9:13] if with_ll else None, scale_ll=8.0)
        out = (yh * gyh.to(dev)).sum()
        if with_ll:
            out = out + (yl * gyl.to(dev)).sum()
        out.backward()
        got[new_path] = [v.grad for v in d]
        for a_, b_, k in zip(d, ref, range(len(ref))):
            assert_close(a_.grad, b_.grad, GRAD_TOL, "head backward (own kernels: %s): leaf %d" % (new_path, k))
    for a_, b_, k in zip(got[True], got[False], range(len(ref))):
        assert_close(a_, b_.cpu(), GRAD_TOL, "own kernels vs generic path: leaf %d" % k)


def test_kitti_decoder_per_head_training_path_still_matches_reference_gradients(dev):
    """dec.stack_heads = False keeps the round-1 per-head operators (with the new gating); same reference gradients."""
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    g = load_golden("kitti_dense_r18_64x64_grads.npz")
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev)
    dec.stack_heads = False
    feats = [f.to(dev).requires_grad_(True) for f in kitti_feats(2, 64, 64)]
    out = dec(feats)
    sum(out[("disp", s)].mean() for s in range(4)).backward()
    for k, f in enumerate(feats):
        assert_close(f.grad, g["dfeat%d" % k], 1e-4, "dfeat%d" % k)
    for name, p in dec.named_parameters():
        assert_close(sample(p.grad.cpu().numpy()), g["d|" + name], 1e-4, name)


def test_a_hook_that_consumes_a_trunk_activation_gets_ungated_gradients(dev):
    """Round-2 ADVICE: the decoders gate every trunk activation's derivative in its consumers and hand the producer dz.  A
    forward hook that uses such an activation in its own loss term is a consumer that does not gate: with a hook registered
    the hints are dropped (layers.gated_backward_allowed), and the gradients equal autograd through the oracle."""
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    from wavelet_monodepth_amd.layers import gated_backward_allowed
    dec = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=1).to(dev)
    assert gated_backward_allowed(dec)
    taken = {}
    h = dec.convs[("upconv", 2, 1)].register_forward_hook(lambda m, i, o: taken.__setitem__("x", o))
    assert not gated_backward_allowed(dec)
    feats = kitti_feats(2, 64, 64)
    gf = [f.to(dev).requires_grad_(True) for f in feats]
    out = dec(gf)
    (sum(out[("disp", s)].mean() for s in range(4)) + 0.01 * (taken["x"] ** 2).mean()).backward()
    h.remove()
    assert gated_backward_allowed(dec)
    # oracle: the same loss through torch.autograd on the CPU restatement
    sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    cf = [f.clone().requires_grad_(True) for f in feats]
    acts = {}
    ref = R.kitti_wave_decoder(cf, sd, activations=acts)
    (sum(ref[("disp", s)].mean() for s in range(4)) + 0.01 * (acts[("upconv", 2, 1)] ** 2).mean()).backward()
    for k in range(5):
        assert_close(gf[k].grad, cf[k].grad, 1e-4, "dfeat%d with a hooked activation" % k)


@pytest.mark.parametrize("shapes", [
    [(32, 48, 3), (16, 16, 1), (7, 5, 3), (40, 21, 1), (3, 64, 3)],
    [(64, 96, 3)] * 3 + [(17, 33, 3), (33, 17, 1)] + [(16, 32, 1)] * 45,      # more filters than one launch holds
])
def test_pack_many_writes_the_images_of_the_single_filter_entry_points(dev, shapes):
    """ops.prepack: one launch for all filters; forward / data-gradient / Winograd images bit-identical to
    wmd_conv_pack_weights[_dgrad] and wmd_conv_pack_weights_wino, and memoised on the tensors."""
    from wavelet_monodepth_amd import ops
    ws = [torch.from_numpy(synth.uniform((co, ci, k, k), "pm", i, -1.0, 1.0)).to(dev) for i, (co, ci, k) in enumerate(shapes)]
    refs = [torch.from_numpy(synth.uniform((co, ci, k, k), "pm", i, -1.0, 1.0)).to(dev) for i, (co, ci, k) in enumerate(shapes)]
    ops.prepack(ws)
    for w, r in zip(ws, refs):
        assert getattr(w, "_wmd_pack_f")[0][0] == w._version
        for dgrad in (False, True):
            got, want = ops.pack_weights(w, dgrad=dgrad), ops.pack_weights(r, dgrad=dgrad)
            assert got.data_ptr() == getattr(w, "_wmd_pack_d" if dgrad else "_wmd_pack_f")[1].data_ptr()   # memo hit
            assert torch.equal(got, want), (tuple(w.shape), dgrad)
            if w.shape[2] == 3:
                assert torch.equal(ops.pack_weights_wino(w, dgrad=dgrad), ops.pack_weights_wino(r, dgrad=dgrad)), (tuple(w.shape), dgrad)
    # an in-place update invalidates the memo; prepack(dgrad=False) only rebuilds the forward images
    with torch.no_grad():
        ws[0].mul_(2.0)
    ops.prepack(ws[:1], dgrad=False)
    assert getattr(ws[0], "_wmd_pack_f")[0][0] == ws[0]._version != getattr(ws[0], "_wmd_pack_d")[0][0]
    assert torch.equal(ops.pack_weights(ws[0]), 2.0 * ops.pack_weights(refs[0]))
