"""GPU parity of the threshold-gated sparse path: mask primitives (bit-exact), the sparse decoder against the
REFERENCE's outputs (tests/golden/kitti_sparse_*.npz: maps, the five mask families, total_ops integers)."""
import numpy as np
import pytest
import torch

from oracle import decoder_ref as R
from wavelet_monodepth_amd import synth
from util import R18, assert_close, key_str, kitti_feats, load_golden, t

pytestmark = pytest.mark.gpu
NET_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_minmax_threshold_dilate_compact_bit_exact(dev):
    from wavelet_monodepth_amd import sparse_ops as S
    for (h, w) in [(3, 5), (12, 40), (96, 320), (1, 1)]:
        yl = t(synth.normal((1, 1, 2 * h, 2 * w), "syl", 5))
        yh = t(synth.normal((1, 1, 3, h, w), "syh", 5))
        mm = S.minmax(yl.to(dev))
        assert float(mm[0]) == float(yl.min()) and float(mm[1]) == float(yl.max())
        for ratio in (0.05, 0.3, -1.0):
            mask = S.mask_threshold(yh.to(dev), mm, ratio)
            ref = (yh.abs().max(2)[0] > (yl.max() - yl.min()) * ratio)[0, 0]
            assert torch.equal(mask.cpu().bool(), ref)
        mask = S.mask_threshold(yh.to(dev), mm, 0.3)
        mf = mask.cpu().float().reshape(1, 1, h, w)
        outs = S.dilate_multi(mask, [(1, 1), (1, 2), (2, 2), (2, 1), (2, 0)])
        refs = [R.dilate(mf, 3), R.dilate(mf, 5), R.dilate(R.up2(mf), 5), R.dilate(R.up2(mf), 3), R.up2(mf)]
        for o, r in zip(outs, refs):
            assert torch.equal(o.cpu().float(), r[0, 0])
        coords, nnz = S.compact_multi(outs)
        for o, c, n in zip(outs, coords, nnz.tolist()):
            ref_idx = torch.nonzero(o.cpu().reshape(-1)).reshape(-1)
            assert n == ref_idx.numel()
            assert torch.equal(c[:n].cpu().long(), ref_idx)          # raster order, like mask2idxmap


def test_fused_mask_level_equals_the_three_primitives_and_the_oracle(dev):
    """wmd_mask_level (min/max + threshold + all dilations, one launch) against the separate entry points (bit-exact) and
    against the oracle's max-pool dilations of the oracle's threshold mask."""
    from wavelet_monodepth_amd import sparse_ops as S
    specs = [(1, 0), (1, 1), (1, 2), (2, 2), (2, 1), (2, 0), (2, 3), (1, 3)]
    for (h, w) in [(1, 1), (3, 5), (7, 9), (12, 40), (48, 160)]:
        yl = t(synth.normal((1, 1, h, w), "fyl", 6))
        yh = t(synth.normal((1, 1, 3, h, w), "fyh", 6))
        for ratio in (0.02, 0.3, 0.6, -1.0, 5.0):
            fused = S.mask_level(yl.to(dev), yh.to(dev), ratio, specs)
            mask = S.mask_threshold(yh.to(dev), S.minmax(yl.to(dev)), ratio)
            sep = S.dilate_multi(mask, specs)
            assert torch.equal(fused[0], mask)
            for f, o in zip(fused, sep):
                assert torch.equal(f, o)
            ref = (yh.abs().max(2)[0] > (yl.max() - yl.min()) * ratio).float()
            assert torch.equal(fused[0].cpu().float(), ref[0, 0])
            assert torch.equal(fused[3].cpu().float(), R.dilate(R.up2(ref), 5)[0, 0])
            assert torch.equal(fused[2].cpu().float(), R.dilate(ref, 5)[0, 0])


@pytest.mark.parametrize("B,h,w", [(1, 12, 40), (3, 24, 80), (12, 48, 160), (2, 7, 9)])
def test_mask_level_lists_masks_lists_counts_and_protocol(dev, B, h, w):
    """wmd_mask_level_lists (round 4): the masks are those of wmd_mask_level_b (bit-exact), the per-frame work lists hold exactly
    the tiles that contain a set pixel (as sets: the order inside a list is unspecified), the pixel counts arrive in the ring
    slot of the forward, stamped; the range keys are consumed and re-armed; scratch is all zero again after every launch;
    three forwards in a row land in three slots (the second and third with an injected mask and an AND mask)."""
    from wavelet_monodepth_amd import sparse_ops as S
    yl = t(synth.normal((B, 1, 2 * h, 2 * w), "lyl", 9)).to(dev)
    yh = t(synth.normal((B, 1, 3, h, w), "lyh", 9)).to(dev)
    st = S.LevelState(dev, B, 2, 16)
    # the range the head epilogue would have left: order-preserving keys of (min, max) per frame
    lo, hi = yl.reshape(B, -1).min(1)[0], yl.reshape(B, -1).max(1)[0]
    key = lambda v: torch.where(v.view(torch.int32) < 0, ~v.view(torch.int32), v.view(torch.int32) | -2147483648)
    st.keys.copy_(torch.stack([key(lo.contiguous()), key(hi.contiguous())], 1))
    specs = [(1, 1, 0, None), (1, 2, 1, (8, 16)), (2, 2, 0, None), (2, 1, 2, (16, 16)), (2, 0, 3, None)]
    ref = S.mask_level(yl, yh, 0.3, [(u, r) for u, r, _c, _t in specs]) if B > 1 else \
        [m.unsqueeze(0) for m in S.mask_level(yl, yh, 0.3, [(u, r) for u, r, _c, _t in specs])]

    def check(masks, lists, want, k, counts_off):
        for m, r in zip(masks, want):
            assert torch.equal(m.reshape(r.shape), r)
        for j, (m, l) in enumerate(zip(masks, lists)):
            if l is None:
                continue
            tl, tc, th, tw = l
            H, W = m.shape[-2:]
            ty, tx = -(-H // th), -(-W // tw)
            pad = torch.zeros((B, ty * th, tx * tw), dtype=torch.uint8)
            pad[:, :H, :W] = m.cpu()
            act = pad.view(B, ty, th, tx, tw).amax((2, 4)).reshape(B, -1)
            cnt = tc.cpu().tolist()
            for f in range(B):
                expect = {int(f * ty * tx + i) for i in torch.nonzero(act[f]).reshape(-1)}
                got = tl.cpu()[f * ty * tx:f * ty * tx + cnt[f]].tolist()
                assert cnt[f] == len(expect) and set(got) == expect and len(set(got)) == len(got), (j, f)
        slot = st.ring.cpu()[(k % st.RING) * st.slot_ints:(k % st.RING + 1) * st.slot_ints].tolist()
        assert slot[0] == k + 1, "stamp"
        for f in range(B):
            got = slot[1 + counts_off + 3 * f:1 + counts_off + 3 * f + 3]
            assert got == [int(masks[1][f].sum()), int(masks[3][f].sum()), int(masks[4][f].sum())], (f, got)
        assert int(st.scratch.abs().sum()) - int(st.scratch[1]) == 0 and int(st.scratch[1]) == k + 1     # all re-armed, seq advanced

    masks, lists = S.mask_level_lists(st, B, h, w, specs, 0.3, yl=yl, yh=yh, use_keys=True, counts_off=0, advance=True)
    check(masks, lists, ref, 0, 0)
    assert st.keys.cpu().tolist() == [[-1, 0]] * B               # consumed and re-armed
    # injected base mask + AND mask (the input support of upconv(i,0)); second counts column block of the slot
    gen = torch.Generator().manual_seed(3)
    m0 = (torch.rand((B, h, w), generator=gen) < 0.2).to(torch.uint8).to(dev)
    andm = (torch.rand((B, h, w), generator=gen) < 0.5).to(torch.uint8).to(dev)
    want = S.dilate_multi(m0, [(u, r) for u, r, _c, _t in specs])
    for k in (1, 2):
        masks, lists = S.mask_level_lists(st, B, h, w, specs + [(1, 1, 0, None, andm)], mask0=m0, counts_off=3 * B, advance=True)
        check(masks[:5], lists[:5], want, k, 3 * B)
        assert torch.equal(masks[5], want[0] & andm)


@pytest.mark.parametrize("case", [(1, 24, 80, 128, 1, 0, 64, (8, 16)), (1, 48, 160, 64, 2, 64, 64, (8, 16)), (12, 24, 80, 32, 2, 32, 32, (16, 16)),
                                  (2, 12, 40, 256, 1, 0, 128, (8, 16)), (1, 22, 50, 16, 1, 0, 40, (8, 16))],
                         ids=lambda c: "x".join(str(v) for v in c[:7]))
@pytest.mark.parametrize("density", [0.0, 0.03, 0.5, 1.0])
def test_conv_work_list_form_equals_the_mask_form_and_the_oracle(dev, case, density):
    """wmd_conv_fwd with out_tiles (round 4: LIST instantiations, device-chosen K split, list-driven second pass) against the
    mask form of the same call (per-block mask tests, host-chosen split) -- identical support, values to 2e-6 -- and against the
    oracle's masked convolution; empty lists, a few tiles (every slice of a deep K split), all tiles, a ragged map (the scalar
    path of the second pass), several frames."""
    from wavelet_monodepth_amd import ops, sparse_ops as S
    B, H, W, C1, up, C2, Cout, tile = case
    x1 = t(synth.normal((B, C1, H // up, W // up), "wlx", 4)).to(dev)
    x2 = t(synth.normal((B, C2, H, W), "wls", 4)).to(dev) if C2 else None
    wgt, bias = [t(a).to(dev) for a in synth.conv_params("wlw", Cout, C1 + C2, 3, 4)]
    gen = torch.Generator().manual_seed(11)
    base = (torch.rand((B, H // 2, W // 2), generator=gen) < density).to(torch.uint8).to(dev)
    st = S.LevelState(dev, B, 1, 16)
    (in_mask, out_mask), lists = S.mask_level_lists(st, B, H // 2, W // 2, [(2, 2, 0, None), (2, 1, 1, tile)], mask0=base, advance=True)
    args = (x1, x2, ops.pack_weights(wgt), bias, Cout, 3, "reflect", "elu", 0.0, up, ops.pack_weights_wino(wgt))
    y_list = torch.full((B, Cout, H, W), 7.0, device=dev)
    ops._conv_fwd_raw(*args, in_mask=in_mask, out_mask=out_mask, out=y_list, in_mask_2x2=True, out_tiles=lists[1])
    y_mask = torch.full((B, Cout, H, W), 7.0, device=dev)
    ops._conv_fwd_raw(*args, in_mask=in_mask, out_mask=out_mask, out=y_mask, in_mask_2x2=True)
    th, tw = tile
    act = torch.zeros((B, -(-H // th) * th, -(-W // tw) * tw), dtype=torch.bool, device=dev)
    act[:, :H, :W] = out_mask.bool()
    act = act.view(B, -1, th, act.shape[2] // tw, tw).amax((2, 4), keepdim=True).expand(-1, -1, th, -1, tw).reshape(B, -1, act.shape[2])[:, :H, :W]
    # unlisted tiles are not written (the never-refilled pool), listed tiles hold the masked convolution
    assert bool((y_list[(~act).unsqueeze(1).expand_as(y_list)] == 7.0).all())
    xin = torch.cat([R.up2(x1.cpu()) if up == 2 else x1.cpu()] + ([x2.cpu()] if C2 else []), 1) * in_mask.cpu().unsqueeze(1).float()
    # (the mask test follows the coordinate padding: pad the masked input, KITTI/layers.py:439-453)
    ref = torch.nn.functional.elu(R.conv3x3(xin, wgt.cpu(), bias.cpu(), "reflect")) * out_mask.cpu().unsqueeze(1).float()
    sel = act.unsqueeze(1).expand_as(y_list).cpu()
    if bool(sel.any()):
        assert_close(y_list.cpu()[sel], ref[sel], 2e-5, "work-list form vs oracle")
        # the mask form computes whole 8x32 / 6x40 ... tiles of its own: compare where both wrote
        both = sel & (y_mask.cpu() != 7.0)
        assert_close(y_list.cpu()[both], y_mask.cpu()[both], 2e-6, "work-list form vs mask form")


@pytest.mark.parametrize("ksize,dual", [(3, False), (1, False), (3, True)])
def test_sparse_conv_every_k_split_mode_vs_dense_reference(dev, ksize, dual):
    """The wavefronts of a block either split K for one tile or take whole K ranges of separate tiles, decided on the
    device from the pixel count (args.split_waves moves the switch point): every mode, with fused upsample + concat,
    reflect index padding, channel counts that are not multiples of 4/16, against a masked dense convolution."""
    import torch.nn.functional as F
    from wavelet_monodepth_amd import ops, sparse_ops as S
    H, W = 40, 56
    c1, c2, cout = (37, 21, 19) if not dual else (24, 0, 3)
    up = 2 if c2 else 1
    gen = torch.Generator().manual_seed(11)
    x1 = torch.randn(2 * c1 if dual else c1, H // up, W // up, generator=gen)
    x2 = torch.randn(c2, H, W, generator=gen) if c2 else None
    in_mask = (torch.rand(H, W, generator=gen) < 0.7)
    omask = (torch.rand(H, W, generator=gen) < 0.5)
    cin = c1 + c2
    w = torch.randn(cout, cin, ksize, ksize, generator=gen) / np.sqrt(cin * ksize * ksize)
    b = torch.randn(cout, generator=gen) * 0.1
    w2 = torch.randn(cout, cin, ksize, ksize, generator=gen) / np.sqrt(cin * ksize * ksize)
    b2 = torch.randn(cout, generator=gen) * 0.1

    def dense(xa, ww, bb):
        xin = xa if up == 1 else F.interpolate(xa[None], scale_factor=2, mode="nearest")[0]
        if x2 is not None:
            xin = torch.cat([xin, x2], 0)
        xin = (xin * (in_mask if ksize == 3 else torch.ones_like(in_mask)))[None]
        if ksize == 3:
            xin = F.pad(xin, (1, 1, 1, 1), mode="reflect")
        return F.elu(F.conv2d(xin, ww, bb))[0]

    ref = dense(x1[:c1], w, b) - dense(x1[c1:], w2, b2) if dual else dense(x1, w, b)
    (coords,), cnt = S.compact_multi([omask.to(dev).to(torch.uint8)])
    ntile = (int(omask.sum()) + 15) // 16
    outs = []
    for sw in (1, ntile + 1, 2 * ntile + 1, 4 * ntile + 1, 2 ** 30):      # -> 1, 2, 4, 8 (and 8) K slices per tile
        y = torch.zeros((cout, H, W), device=dev)
        S.sparse_conv(y, x1.to(dev), ops.pack_weights(w.to(dev)), b.to(dev), cout, ksize, coords, cnt.data_ptr(), H * W,
                      x2=None if x2 is None else x2.to(dev), up1=up, in_mask=in_mask.to(dev).to(torch.uint8) if ksize == 3 else None,
                      pad="reflect", act="elu", split_waves=sw,
                      **(dict(c1=c1, c1_off=0, wp2=ops.pack_weights(w2.to(dev)), bias2=b2.to(dev), c1_off2=c1) if dual else {}))
        y = y.cpu()
        assert float(y[:, ~omask].abs().max()) == 0.0, "wrote outside the output mask"
        assert_close(y[:, omask], ref[:, omask], 2e-5, "split_waves=%d" % sw)
        outs.append(y)
    for y in outs[1:]:
        assert_close(y, outs[0], 2e-6, "K-split modes agree")


def test_sparse_conv_matches_reference_primitive_golden(dev):
    """sparse_conv3x3 of the reference (KITTI/layers.py:409-480) on a hand-made mask pair, both index paddings."""
    from wavelet_monodepth_amd import ops, sparse_ops as S
    g = load_golden("kitti_sparse_primitives.npz")
    mask, omask = t(g["mask"])[0, 0], t(g["omask"])[0, 0]
    cin, cout = 5, 4
    nnz = int(mask.sum())
    vals = t(synth.normal((cin * nnz,), "pv", 5)).reshape(cin, nnz)
    dense_in = R.scatter_dense(vals, mask)[0].to(dev)                  # dense zero-initialised layout
    bound = 1.0 / np.sqrt(cin * 9)
    w = t(synth.uniform((cout, cin, 3, 3), "conv.weight", 5, -bound, bound)).to(dev)
    b = t(synth.uniform((cout,), "conv.bias", 5, -bound, bound)).to(dev)
    (coords,), cnt = S.compact_multi([omask.to(dev).to(torch.uint8)])
    for pad, name in (("reflect", "reflect"), ("zero", "constant")):
        y = torch.zeros((cout, 6, 9), device=dev)
        S.sparse_conv(y, dense_in, ops.pack_weights(w), b, cout, 3, coords, cnt.data_ptr(), 54,
                      in_mask=mask.to(dev).to(torch.uint8), pad=pad)
        assert_close(y.unsqueeze(0), g["sconv_dense_" + name], 2e-5, "sparse conv " + name)


def _decoder(dev, seed=1):
    from wavelet_monodepth_amd.kitti import SparseDepthWaveProgressiveDecoder
    return synth.fill_state_dict(SparseDepthWaveProgressiveDecoder(np.array(R18)), seed=seed).to(dev)


def _check(out, gold, exact_masks=True, tol=NET_TOL):
    assert set(key_str(k) for k in out) == set(gold), set(gold) ^ set(key_str(k) for k in out)
    for k, v in out.items():
        ks = key_str(k)
        if torch.is_tensor(v) and v.dtype == torch.bool:
            if exact_masks:
                assert np.array_equal(v.cpu().numpy().astype(np.uint8), gold[ks]), ks
        elif torch.is_tensor(v):
            assert_close(v, gold[ks], tol, ks)
        elif exact_masks:
            assert int(v) == int(gold[ks]), "%s: %d vs %d" % (ks, int(v), int(gold[ks]))


@pytest.mark.parametrize("thr", [-1.0, 0.01])
def test_sparse_decoder_full_density_vs_reference_golden(dev, thr):
    gold = load_golden("kitti_sparse_r18_64x64_thr%g.npz" % thr)
    out = _decoder(dev)([f[:1].to(dev) for f in kitti_feats(2, 64, 64)], thr)
    _check(out, gold)


@pytest.mark.parametrize("tiles", ["0", "1"], ids=["gather", "tiles"])
@pytest.mark.parametrize("name,hw,seed,thr", [("64x64", (64, 64), 1, 0.05), ("64x64", (64, 64), 1, 0.1),
                                              ("96x160", (96, 160), 2, 0.15), ("96x160", (96, 160), 2, 0.2)])
def test_sparse_decoder_vs_reference_golden_with_reference_masks(dev, name, hw, seed, thr, tiles, monkeypatch):
    """Feed the reference's own threshold masks (a pixel sitting exactly at the threshold may legitimately
    flip with fp32 rounding); everything downstream — dilations, compaction, gather-GEMMs, heads, IDWT, the five
    mask families and the integer op model — must then match the reference exactly / to 1e-4.  Both forms of the sparse
    levels: gather-GEMMs over the pixel lists, and the block-sparse dense kernels + masked fused heads."""
    monkeypatch.setenv("WMD_SPARSE_TILES", tiles)
    gold = load_golden("kitti_sparse_r18_%s_thr%g.npz" % (name, thr))
    feats = kitti_feats(2 if name == "64x64" else 1, hw[0], hw[1], seed=seed)
    force = {i: t(gold["wavelet_mask|%d" % (i - 1)])[0, 0, ::2, ::2] for i in (3, 2, 1)}
    out = _decoder(dev)([f[:1].to(dev) for f in feats], thr, _force_masks=force)
    _check(out, gold)


def test_sparse_decoder_empty_masks_vs_reference_golden(dev):
    """thresh_ratio above 1: no coefficient passes, every compacted list has nnz = 0 (the reference then runs its
    gather-GEMMs on empty tensors); masks, op count and maps must still match, eagerly and from the replayed graph."""
    gold = load_golden("kitti_sparse_r18_64x64_thr2.npz")
    sp = _decoder(dev)
    feats = [f[:1].to(dev) for f in kitti_feats(2, 64, 64)]
    _check(sp(feats, 2.0), gold)
    sp.enable_graph(True)
    for _ in range(2):
        out = sp(feats, 2.0)
    _check(out, gold)


@pytest.mark.parametrize("name,hw,seed,thr", [("64x64", (64, 64), 1, 0.1), ("96x160", (96, 160), 2, 0.15)])
def test_sparse_decoder_free_running_masks(dev, name, hw, seed, thr):
    gold = load_golden("kitti_sparse_r18_%s_thr%g.npz" % (name, thr))
    feats = kitti_feats(2 if name == "64x64" else 1, hw[0], hw[1], seed=seed)
    out = _decoder(dev)([f[:1].to(dev) for f in feats], thr)
    for s in range(3):
        m = out[("wavelet_mask", s)].cpu().numpy().astype(np.uint8)
        assert (m != gold["wavelet_mask|%d" % s]).mean() < 0.005
    # where masks agree the maps agree; a flipped threshold pixel changes a coefficient that is ~thresh itself
    assert float(np.abs(out[("disp", 0)].cpu().numpy() - gold["disp|0"]).mean()) < 1e-4


def test_sparse_decoder_graph_replay_matches_eager(dev):
    sp = _decoder(dev, seed=3)
    feats = [f.to(dev) for f in kitti_feats(1, 96, 160, seed=2)]
    ref = sp(feats, 0.15)
    sp.enable_graph(True)
    for _ in range(2):
        out = sp(feats, 0.15)
    assert out["total_ops"] == ref["total_ops"]
    for k, v in ref.items():
        if torch.is_tensor(v):
            if v.dtype == torch.bool:
                assert torch.equal(out[k], v), key_str(k)
            else:
                assert_close(out[k], v, 2e-6, key_str(k))


def test_sparse_decoder_graph_replay_with_injected_masks(dev):
    """Injected (device-resident) masks are live inputs of the captured graph: replay == eager, and new mask values
    written in place flow through."""
    sp = _decoder(dev, seed=3)
    feats = [f.to(dev) for f in kitti_feats(1, 96, 160, seed=2)]
    shapes = {3: (6, 10), 2: (12, 20), 1: (24, 40)}
    gen = torch.Generator().manual_seed(5)
    force = {i: (torch.rand(hw, generator=gen) < 0.3).to(torch.uint8).to(dev) for i, hw in shapes.items()}
    ref = sp(feats, 0.05, _force_masks={i: m.clone() for i, m in force.items()})
    sp.enable_graph(True)
    for _ in range(2):
        out = sp(feats, 0.05, _force_masks=force)
    assert out["total_ops"] == ref["total_ops"]
    for s_ in range(4):
        assert_close(out[("disp", s_)], ref[("disp", s_)], 2e-6, "disp%d" % s_)
    sp.enable_graph(False)
    for i in force:
        force[i].copy_((torch.rand(shapes[i], generator=gen) < 0.6).to(torch.uint8))
    ref2 = sp(feats, 0.05, _force_masks={i: m.clone() for i, m in force.items()})
    sp.enable_graph(True)
    out2 = sp(feats, 0.05, _force_masks=force)
    assert out2["total_ops"] == ref2["total_ops"] and out2["total_ops"] != ref["total_ops"]
    for s_ in range(4):
        assert_close(out2[("disp", s_)], ref2[("disp", s_)], 2e-6, "disp%d after in-place mask update" % s_)


def test_total_ops_of_an_old_forward_survives_the_count_ring(dev):
    """The work-list form publishes a forward's pixel counts into slot k % 64 of a device-side ring and reads them only when
    `total_ops` is asked for.  An output dictionary that is still unresolved when its slot comes round again is resolved by the
    forward that is about to reuse the slot: 70 forwards with alternating thresholds, the first two outputs read last."""
    sp = _decoder(dev, seed=3)
    feats = [f.to(dev) for f in kitti_feats(1, 96, 160, seed=2)]
    want = {thr: sp(feats, thr)["total_ops"] for thr in (0.15, 0.3)}
    assert want[0.15] != want[0.3]
    for graph in (False, True):
        sp.enable_graph(graph)
        outs = [sp(feats, (0.15, 0.3)[k & 1]) for k in range(70)]
        assert outs[0]["total_ops"] == want[0.15] and outs[1]["total_ops"] == want[0.3]
        assert outs[69]["total_ops"] == want[0.3] and outs[64]["total_ops"] == want[0.15]
    sp.enable_graph(False)


def test_sparse_equals_dense_at_negative_threshold(dev):
    """Reference invariant: thresh_ratio <= 0 reproduces the dense decoder with the same weights."""
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    sp = _decoder(dev, seed=7)
    dn = DepthWaveProgressiveDecoder(np.array(R18)).to(dev)
    dn.load_state_dict(sp.state_dict())
    feats = [f.to(dev) for f in kitti_feats(1, 192, 640, seed=7)]
    with torch.no_grad():
        a, b = sp(feats, -1.0), dn(feats)
    for s in range(4):
        assert_close(a[("disp", s)], b[("disp", s)], 2e-5, "disp%d" % s)
    assert a["total_ops"] == 3560015775      # notebook known answer for R18 640x192


NYU_ENC = [8, 8, 16, 32, 64]


def _nyu(dev):
    from wavelet_monodepth_amd.nyu import SparseDecoderWave
    return synth.fill_state_dict(SparseDecoderWave(enc_features=NYU_ENC), seed=8).to(dev)


def _check_nyu(out, gold, tol=NET_TOL):
    assert set(key_str(k) for k in out) == set(gold), set(gold) ^ set(key_str(k) for k in out)
    for k, v in out.items():
        ks = key_str(k)
        if torch.is_tensor(v):
            assert_close(v, gold[ks], tol, ks)
        else:
            assert int(v) == int(gold[ks]), "%s: %d vs %d" % (ks, int(v), int(gold[ks]))


@pytest.mark.parametrize("thr", [-1.0, 0.02])
def test_nyu_sparse_decoder_full_density_vs_reference_golden(dev, thr):
    from util import nyu_feats
    gold = load_golden("nyu_sparse_small_64x96_thr%g.npz" % thr)
    out = _nyu(dev)([f[:1].to(dev) for f in nyu_feats(2, 64, 96, NYU_ENC)], thr)
    _check_nyu(out, gold)


def test_nyu_sparse_decoder_vs_reference_golden_with_reference_masks(dev):
    from util import nyu_feats
    gold = load_golden("nyu_sparse_small_64x96_thr0.1.npz")
    dens = [float(gold["wavelet_mask|%d" % s].mean()) for s in (1, 0)]
    assert min(dens) < 0.9, "fixture should exercise a non-trivial mask: %s" % dens
    force = {lvl: t(gold["wavelet_mask|%d" % (1 - lvl)])[0, 0, ::2, ::2] for lvl in (0, 1)}
    out = _nyu(dev)([f[:1].to(dev) for f in nyu_feats(2, 64, 96, NYU_ENC)], 0.1, _force_masks=force)
    _check_nyu(out, gold)


@pytest.mark.parametrize("thr,forced", [(0.15, False), (0.05, True), (2.0, False)])
def test_sparse_decoder_batched_equals_per_frame(dev, thr, forced, monkeypatch):
    """Extension over the reference's batch-1 assert (depth_decoder.py:297): B frames decoded through the same launches,
    each with its own coefficient range, masks, pixel lists and counts.  Every per-frame output -- maps, the five mask
    families, the integer op model -- must equal the batch-1 decode of that frame; eager and from the replayed graph."""
    sp = _decoder(dev, seed=3)
    B = 3
    frames = [[f.to(dev) for f in kitti_feats(1, 96, 160, seed=20 + k)] for k in range(B)]
    batch = [torch.cat([fr[j] for fr in frames], 0) for j in range(5)]
    force_b = force_k = None
    if forced:
        gen = torch.Generator().manual_seed(7)
        shapes = {3: (6, 10), 2: (12, 20), 1: (24, 40)}
        force_b = {i: (torch.rand((B,) + hw, generator=gen) < 0.25).to(torch.uint8).to(dev) for i, hw in shapes.items()}
        force_k = [{i: m[k].clone() for i, m in force_b.items()} for k in range(B)]
    monkeypatch.setenv("WMD_SPARSE_TILES", "0")       # the per-frame references run the gather-GEMM form ...
    singles = [dict(sp(frames[k], thr, _force_masks=None if not forced else force_k[k]).items()) for k in range(B)]
    monkeypatch.delenv("WMD_SPARSE_TILES")            # ... the batch the block-sparse dense kernels
    for graph in (False, True):
        sp.enable_graph(graph)
        for _ in range(2 if graph else 1):
            out = sp(batch, thr, _force_masks=force_b)
        assert isinstance(out["total_ops"], list) and len(out["total_ops"]) == B
        for k in range(B):
            for key, ref in singles[k].items():
                v = out[key]
                if torch.is_tensor(ref):
                    if ref.dtype == torch.bool:
                        assert torch.equal(v[k:k + 1], ref), "frame %d %s" % (k, key_str(key))
                    else:
                        # (a batch runs the trunk on the block-sparse Winograd kernels, one frame on the gather-GEMM)
                        assert_close(v[k:k + 1], ref, 1e-5, "frame %d %s" % (k, key_str(key)))
                else:
                    assert v[k] == ref, "frame %d %s: %s vs %s" % (k, key_str(key), v[k], ref)
    sp.enable_graph(False)
