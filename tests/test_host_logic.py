"""Host-side logic that needs no GPU: the sparse decoders' op model (the python-int `total_ops` the reference returns)
rebuilt from the REFERENCE's own masks and checked against the reference's own counts (tests/golden/kitti_sparse_*.npz),
and the pooled-buffer arithmetic."""
import numpy as np
import pytest

from util import R18, load_golden
from wavelet_monodepth_amd.kitti import sparse_decoder as SD

DEC = [16, 32, 64, 128, 256]


def _kitti_total_ops(gold, sparse_levels=(3, 2, 1)):
    static_ops, counters, counts = {}, [], {}
    for i in (4, 3, 2, 1):
        h, w = gold["lowres_mask|%d" % (i - 1)].shape[-2:]
        C = DEC[i]
        trunk = ((R18[4] if i == 4 else DEC[i + 1], C), (C + R18[i - 1], C))
        if i in sparse_levels:
            static_ops[i] = SD.level_static_ops(i, h, w, True)
            counters.append((i, None, trunk[0], trunk[1], (C, C), (C, 3)))
            counts[i] = tuple(int(gold["%s_mask|%d" % (m, i - 1)].sum()) for m in ("upconv0", "upconv1", "wavelet"))
        else:
            heads = ([(C, C // 4, C // 4, 1)] if i == 4 else []) + [(C, C, C, 3), (C, C, C, 3)]
            static_ops[i] = SD.level_static_ops(i, h, w, False, trunk, heads)
    return SD.resolve_total_ops(static_ops, counters, counts)


@pytest.mark.parametrize("name", ["64x64_thr-1", "64x64_thr0.01", "64x64_thr0.05", "64x64_thr0.1", "64x64_thr2",
                                  "96x160_thr0.15", "96x160_thr0.2"])
def test_kitti_sparse_op_model_from_reference_masks(name):
    gold = load_golden("kitti_sparse_r18_%s.npz" % name)
    per_scale, total = _kitti_total_ops(gold)
    for s in range(4):
        assert per_scale[s] == int(gold["total_ops|%d" % s]), "scale %d" % s
    assert total == int(gold["total_ops"])


@pytest.mark.parametrize("name", ["64x64_thr0.05", "64x64_thr0.1", "96x160_thr0.15", "96x160_thr0.2"])
def test_reference_upsample_mask_is_the_upsampled_lowres_mask(name):
    """The block-sparse form of the second trunk convolution promises the kernel that its input mask is constant on 2x2
    blocks (wmd_conv_args.in_mask_2x2): MaxPool2d(5)(upsample(m)) == upsample(MaxPool2d(3)(m)) (depth_decoder.py:311-316).
    Checked on the masks the reference itself produced."""
    gold = load_golden("kitti_sparse_r18_%s.npz" % name)
    for s in range(4):
        lo, up = gold["lowres_mask|%d" % s], gold["upsample_mask|%d" % s]
        assert np.array_equal(lo.repeat(2, -2).repeat(2, -1), up), "scale %d" % s


def test_kitti_sparse_op_model_full_size_known_answer():
    """R18 640x192 with every pixel active: the notebook's printed 3.560 G."""
    gold = {}
    for i, (h, w) in zip((4, 3, 2, 1), ((6, 20), (12, 40), (24, 80), (48, 160))):
        gold["lowres_mask|%d" % (i - 1)] = np.ones((1, 1, h, w), np.uint8)
        gold["upconv0_mask|%d" % (i - 1)] = np.ones((1, 1, h, w), np.uint8)
        gold["upconv1_mask|%d" % (i - 1)] = np.ones((1, 1, 2 * h, 2 * w), np.uint8)
        gold["wavelet_mask|%d" % (i - 1)] = np.ones((1, 1, 2 * h, 2 * w), np.uint8)
    assert _kitti_total_ops(gold)[1] == 3560015775


def test_pool_rounding():
    assert [SD._round64(n) for n in (0, 1, 63, 64, 65, 128)] == [0, 64, 64, 64, 128, 128]


# ---- NYUv2 SparseDecoderWave -------------------------------------------------------------------------------------------
def _nyu_total_ops(enc, hw_in, counts):
    from wavelet_monodepth_amd.nyu.densedepth_decoder import nyu_sparse_total_ops
    F = int(enc[-1] * 0.5)
    h0, w0 = hw_in
    return nyu_sparse_total_ops(enc[-1], hw_in, F, enc[-2], F // 2, (2 * h0, 2 * w0),
                                [(F // 2 + enc[-3], F // 4, F // 4), (F // 4 + enc[-4], F // 8, F // 8)], counts)


@pytest.mark.parametrize("thr", ["-1", "0.02", "0.1"])
def test_nyu_sparse_op_model_from_reference_masks(thr):
    import torch
    from oracle import decoder_ref as R
    gold = load_golden("nyu_sparse_small_64x96_thr%s.npz" % thr)
    counts = []
    for s in (1, 0):
        wl = torch.from_numpy(gold["wavelet_mask|%d" % s]).float()
        counts.append((int(R.dilate(wl, 3).sum()), int(wl.sum())))      # wave mask = 3x3 dilation of the wavelet mask
    assert _nyu_total_ops([8, 8, 16, 32, 64], (2, 3), counts) == int(gold["total_ops"])


def test_nyu_sparse_op_model_full_size_known_answer():
    """DenseNet161-shaped decoder at 640x480 with every pixel active (tests/golden/nyu_total_ops.json, produced by the
    reference at thresh_ratio = -1)."""
    import json, os
    with open(os.path.join(os.path.dirname(__file__), "golden", "nyu_total_ops.json")) as f:
        want = json.load(f)["nyu_densenet161_640x480"]["total_ops"]
    counts = [(60 * 80, 60 * 80), (120 * 160, 120 * 160)]
    assert _nyu_total_ops([96, 96, 192, 384, 2208], (15, 20), counts) == want


# ---- autotuner (host logic; the configuration table comes from the library, no GPU needed) ---------------------------------
def test_committed_tune_cache_names_exist_in_the_library_table():
    """bench.py preloads profiles/<bench.TUNE_CACHE>; an entry whose kernel name left the table would be ignored silently
    (and the shape re-tuned inside the warm-up).  The forward / data-gradient entries name convolution configurations."""
    import importlib.util, json, os
    from wavelet_monodepth_amd import tuner
    root = os.path.dirname(os.path.dirname(__file__))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    names = tuner.config_names()
    assert len(names) >= 10 and len(set(names)) == len(names)
    with open(os.path.join(root, "profiles", bench.TUNE_CACHE)) as f:
        cache = json.load(f)
    conv = {k: v for k, v in cache.items() if k.startswith("conv|") or k.startswith("dgrad|")}
    assert len(conv) >= 8 and any("|12|96|320|32|2|64|32|3" in k for k in conv)      # the headline workload's dominant layer
    for key, (name, ks) in conv.items():
        assert name in names, "%s -> %s is not a kernel configuration of this build" % (key, name)
        assert abs(ks) in tuner.KSPLITS      # (-k: k slices summed by the second-stage kernel, round 6)


def test_tuner_picks_the_fastest_valid_candidate(monkeypatch):
    import torch
    from wavelet_monodepth_amd import tuner
    names = tuner.config_names()
    cands = [i + 1 for i, n in enumerate(names) if n.endswith(",9>") or n.startswith("conv_wino")]
    fast, refused, hopeless = cands[3], cands[1], cands[5]
    calls = []

    def launch(cfg, ks):
        calls.append((cfg, ks))
        return -3 if cfg == refused or ks > 4 else 0           # planner refuses: not a candidate

    def fake_time(launch_, cfg, ks, reps, e0, e1):
        if cfg == fast:
            return 1.0 if ks == 2 else 1.5
        return 100.0 if cfg == hopeless else 3.0 + 0.01 * cfg + 0.1 * ks

    class _Ev:
        def __init__(self, enable_timing=True):
            pass

    monkeypatch.setattr(tuner, "_time", fake_time)
    monkeypatch.setattr(torch.cuda, "Event", _Ev)
    monkeypatch.setattr(tuner, "_cache_file", None)
    key = "conv|test|host-logic"
    try:
        cfg, ks = tuner.tune(key, 9, launch)
        assert (cfg, ks) == (fast, 2)
        assert tuner.lookup(key) == (fast, 2)                  # remembered by NAME, resolved back to the index
        assert not any(c == hopeless and k > 1 for c, k in calls), "a hopeless tile shape had its splits swept"
        # a name that is not in the table (stale cache) is ignored, not mis-resolved
        tuner._cache["conv|stale"] = ("conv_fwd_kernel<0,0,0,0,0,0,0,9>", 1)
        assert tuner.lookup("conv|stale") is None
        # preload never overrides what this process measured
        import json, tempfile, os
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump({key: [names[cands[0] - 1], 1], "conv|other": [names[cands[0] - 1], 4]}, f)
        tuner.preload(f.name)
        os.unlink(f.name)
        assert tuner.lookup(key) == (fast, 2) and tuner.lookup("conv|other") == (cands[0], 4)
    finally:
        for k in (key, "conv|stale", "conv|other"):
            tuner._cache.pop(k, None)


# ---- checkpoint format (KITTI/trainer.py:733-751: torch.save(model.state_dict()) per model) -------------------------------
def _modules():
    from wavelet_monodepth_amd import kitti, nyu
    ch = {"r18": [64, 64, 128, 256, 512], "r50": [64, 256, 512, 1024, 2048]}
    mods = {}
    for tag, c in ch.items():
        for cls in (kitti.DepthDecoder, kitti.DepthWaveProgressiveDecoder, kitti.SparseDepthWaveProgressiveDecoder):
            mods[("kitti", "%s|%s" % (cls.__name__, tag))] = lambda cls=cls, c=c: cls(np.array(c))
    for cls, kws in ((nyu.Decoder, [{}, {"is_depthwise": True}]), (nyu.Decoder224, [{}, {"is_depthwise": True}]),
                     (nyu.DecoderWave, [{}, {"dw_waveconv": True, "dw_upconv": True}]), (nyu.DecoderWave224, [{}]),
                     (nyu.SparseDecoderWave, [{}])):
        for kw in kws:
            tag = ",".join("%s=%s" % kv for kv in sorted(kw.items()))
            mods[("nyu", "%s|%s" % (cls.__name__, tag))] = lambda cls=cls, kw=kw: cls(**kw)
    return mods


def test_state_dict_names_and_shapes_are_the_references():
    """Every decoder class, R18/R50 resp. DenseNet161 widths, depthwise options: parameter names and shapes equal the
    reference's (tests/golden/state_dict_manifest_*.json, written by the reference's own classes)."""
    import json, os
    man = {}
    for proj in ("kitti", "nyu"):
        with open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_manifest_%s.json" % proj)) as f:
            man[proj] = json.load(f)
    mods = _modules()
    assert {k[1] for k in mods if k[0] == "kitti"} == set(man["kitti"]) and {k[1] for k in mods if k[0] == "nyu"} == set(man["nyu"])
    wavelet_buffers = {"g0_col", "g1_col", "g0_row", "g1_row"}
    for (proj, name), make in mods.items():
        sd = make().state_dict()
        own = {k: list(v.shape) for k, v in sd.items() if k.split(".")[0] not in ("inverse_wt", "iwt", "iwt_LL")}
        assert own == man[proj][name], "%s: %s" % (name, set(own) ^ set(man[proj][name]))
        for k in sd:                                       # what is left are the upstream DWTInverse filter buffers
            if k.split(".")[0] in ("inverse_wt", "iwt", "iwt_LL"):
                assert k.split(".", 1)[1] in wavelet_buffers


def test_reference_format_checkpoint_roundtrip(tmp_path):
    """A file written the way the reference's trainer writes it (torch.save of a flat state_dict; the encoder file also
    carries height/width/use_stereo, trainer.py:742-747) loads with strict=True; unknown upstream wavelet buffer names are
    tolerated; values survive the round trip bit for bit."""
    import torch
    from wavelet_monodepth_amd import synth
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder
    src = synth.fill_state_dict(DepthWaveProgressiveDecoder(np.array(R18)), seed=3)
    sd = dict(src.state_dict())
    sd["inverse_wt.h0_col"] = torch.zeros(1, 1, 2, 1)          # an upstream buffer name this build does not know
    path = str(tmp_path / "depth.pth")
    torch.save(sd, path)
    for cls in (DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder):   # interchangeable checkpoints (:186-189)
        dst = cls(np.array(R18))
        # the reference's own loading idiom (trainer.py:768-772 / test_simple.py:101-102)
        model_dict = dst.state_dict()
        pretrained = {k: v for k, v in torch.load(path).items() if k in model_dict}
        model_dict.update(pretrained)
        dst.load_state_dict(model_dict)
        dst2 = cls(np.array(R18))
        dst2.load_state_dict(torch.load(path), strict=True)
        for k, v in src.state_dict().items():
            assert torch.equal(dst.state_dict()[k], v) and torch.equal(dst2.state_dict()[k], v), k


def test_eager_context_suspends_graph_mode_and_keeps_the_cache():
    """decoder.eager() (bench.py's per-kernel pass between capture and the timed replays): graph mode is off inside the context and
    back on behind it -- also when the body raises -- and the cache of captured graphs is the same object with the same entries
    (enable_graph(False) would have cleared it)."""
    import pytest
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    dec = DepthWaveProgressiveDecoder(np.array(R18))
    dec.enable_graph(True)
    cache = dec._graphs
    cache._entries["sentinel"] = object()
    with dec.eager() as d:
        assert d is dec and dec._graph_mode is False
        assert dec._graphs is cache and "sentinel" in cache._entries
    assert dec._graph_mode is True and "sentinel" in cache._entries
    with pytest.raises(RuntimeError):
        with dec.eager():
            raise RuntimeError("body failed")
    assert dec._graph_mode is True
    dec.enable_graph(False)
    with dec.eager():
        assert dec._graph_mode is False
    assert dec._graph_mode is False          # an eager decoder stays eager
    assert "sentinel" not in dec._graphs._entries


# ---- no CPU fallback anywhere in the product ----------------------------------------------------------------------------
def test_every_module_refuses_cpu_tensors():
    """The product path must fail loudly instead of computing on the host: decoders, wavelet modules, loss and
    evaluation operators all raise WmdError on CPU tensors (this box has no GPU)."""
    import torch
    from wavelet_monodepth_amd import _lib, evaluation as ev, ops, photometric as ph, sparse_ops as S
    from wavelet_monodepth_amd.kitti import DepthDecoder, DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder
    from wavelet_monodepth_amd.nyu import DecoderWave, SparseDecoderWave
    from wavelet_monodepth_amd.wavelets import DWT, IDWT
    kf = [torch.zeros(1, c, 64 >> (k + 1), 64 >> (k + 1)) for k, c in enumerate(R18)]
    enc = [8, 8, 16, 32, 64]
    nf = [torch.zeros(1, c, 64 >> (k + 1), 96 >> (k + 1)) for k, c in enumerate(enc)]
    calls = [
        lambda: DepthWaveProgressiveDecoder(np.array(R18))(kf),
        lambda: DepthDecoder(np.array(R18))(kf),
        lambda: SparseDepthWaveProgressiveDecoder(np.array(R18))(kf, 0.05),
        lambda: DecoderWave(enc_features=enc)(nf),
        lambda: SparseDecoderWave(enc_features=enc)(nf, 0.1),
        lambda: IDWT()((torch.zeros(1, 1, 4, 4), [torch.zeros(1, 1, 3, 4, 4)])),
        lambda: DWT(J=2)(torch.zeros(1, 1, 8, 8)),
        lambda: ph.SSIM()(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8)),
        lambda: ph.get_smooth_loss(torch.zeros(1, 1, 8, 8), torch.zeros(1, 3, 8, 8)),
        lambda: ev.compute_errors(torch.ones(16), torch.ones(16)),
        lambda: S.minmax(torch.zeros(8)),
        lambda: S.mask_threshold(torch.zeros(1, 1, 3, 4, 4), torch.zeros(2), 0.1),
        lambda: S.dilate_multi(torch.zeros(4, 4, dtype=torch.uint8), [(1, 1)]),
        lambda: S.mask_level(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 3, 4, 4), 0.1, [(1, 0)]),
        lambda: S.compact_multi([torch.zeros(4, 4, dtype=torch.uint8)]),
        lambda: S.sparse_conv(torch.zeros(4, 4, 4), torch.zeros(4, 4, 4), torch.zeros(2304), None, 4, 3,
                              torch.zeros(16, dtype=torch.int32), 0, 16),
        lambda: ops.upsample_bilinear(torch.zeros(1, 1, 4, 4), (8, 8)),
        lambda: ops.dwt_haar(torch.zeros(1, 1, 8, 8), 1),
        lambda: ops.conv2d_fused(torch.zeros(1, 4, 4, 4), torch.zeros(4, 4, 3, 3), None),
        lambda: ops.dwconv3x3_relu(torch.zeros(1, 4, 4, 4), torch.zeros(4, 1, 3, 3)),
        lambda: ev.flip_postprocess(torch.zeros(1, 4, 4), torch.zeros(1, 4, 4)),
        lambda: ev.compute_errors_nyu(torch.ones(1, 16), torch.ones(1, 16)),
        lambda: ph.warp_frame(torch.zeros(1, 3, 8, 8), torch.ones(1, 1, 8, 8), torch.eye(4)[None], torch.eye(4)[None],
                              torch.eye(4)[None]),
    ]
    for i, call in enumerate(calls):
        with pytest.raises((_lib.WmdError, AssertionError)) as e:
            with torch.no_grad():
                call()
        assert isinstance(e.value, _lib.WmdError), "call %d: %r" % (i, e.value)


def test_sparse_entry_points_validate_arguments_without_gpu():
    import ctypes as C
    from wavelet_monodepth_amd import _lib
    lib = _lib.lib()
    spec = (_lib.DilateSpec * 1)(_lib.DilateSpec(1, 0, 1))
    assert lib.wmd_mask_level(None, 4, None, 0.1, 2, 2, spec, 1, None) == -1                      # null planes
    assert lib.wmd_mask_level(1, 0, 1, 0.1, 2, 2, spec, 1, None) == -2                            # empty yl (torch.max raises too)
    assert lib.wmd_mask_level(1, 4, 1, 0.1, 2, 2, spec, 9, None) == -2                            # more than 8 variants
    bad = (_lib.DilateSpec * 1)(_lib.DilateSpec(3, 0, 1))
    assert lib.wmd_mask_level(1, 4, 1, 0.1, 2, 2, bad, 1, None) == -1                             # up must be 1 or 2
    a = _lib.SparseConvArgs(H=4, W=4, C1=4, up1=1, C1tot=4, c1_off=0, C2=0, Cout=4, ksize=3, pad_mode=1, act=0, slope=0.0,
                            x1=1, x2=None, in_mask=None, out_coords=1, out_nnz=1, max_out=16, wp=1, bias=None, wp2=None,
                            bias2=None, c1_off2=0, out_scale=1.0, y=1, split_waves=-1)
    assert lib.wmd_sparse_conv(C.byref(a), None) == -1 and b"split_waves" in lib.wmd_last_error()
    a.split_waves, a.ksize = 0, 5
    assert lib.wmd_sparse_conv(C.byref(a), None) == -3                                            # unsupported kernel size
    a.ksize, a.max_out = 3, 0
    assert lib.wmd_sparse_conv(C.byref(a), None) == 0                                             # nothing to do: no launch


def test_round3_entry_points_validate_arguments_without_gpu():
    """wmd_head3x3_bwd / wmd_head1x1_bwd / wmd_head_bwd and the encoder-edge fields of wmd_conv_args reject bad arguments
    before any launch (status codes of include/wmd.h: -1 bad argument, -2 bad shape, -3 unsupported, -5 workspace)."""
    import ctypes as C
    from wavelet_monodepth_amd import _lib
    lib = _lib.lib()
    h = _lib.HeadBwdHead(row0=0, nrows=3, ch0=0, nch=32, w3=1, dw3=1, db3=1)
    a3 = _lib.Head3x3BwdArgs(B=1, H=8, W=8, Ct=64, n_out=6, pad_mode=1, act=2, slope=0.1, dy3=1, mid=1, dzmid=1, n_heads=1,
                             workspace=None, workspace_floats=0)
    a3.head[0] = h
    assert lib.wmd_head3x3_bwd_workspace_floats(C.byref(a3)) > 0
    assert lib.wmd_head3x3_bwd(C.byref(a3), None) == -5 and b"workspace" in lib.wmd_last_error()
    a3.head[0].nrows = 2
    assert lib.wmd_head3x3_bwd(C.byref(a3), None) == -1                                            # heads have 1 or 3 outputs
    a3.head[0].nrows, a3.head[0].nch = 3, 65
    assert lib.wmd_head3x3_bwd(C.byref(a3), None) == -1                                            # channels beyond Ct
    a3.head[0].nch, a3.act = 32, 3
    assert lib.wmd_head3x3_bwd(C.byref(a3), None) == -3                                            # sigmoid-gated mid: unsupported
    a3.act, a3.H = 2, 1
    assert lib.wmd_head3x3_bwd(C.byref(a3), None) == -2                                            # reflection needs H >= 2
    a3.H = 8
    a1 = _lib.Head1x1BwdArgs(B=1, H=8, W=8, C=32, Ct=64, x_act=1, x_slope=0.0, dz=1, x=1, w1=1, dx=1, dw1=1, db1=1,
                             workspace=None, workspace_floats=0)
    assert lib.wmd_head1x1_bwd_workspace_floats(C.byref(a1)) > 0
    assert lib.wmd_head1x1_bwd(C.byref(a1), None) == -5
    a1.Ct = 60
    assert lib.wmd_head1x1_bwd(C.byref(a1), None) == -3                                            # Ct must be a multiple of 8
    a1.Ct, a1.dz = 64, 2
    assert lib.wmd_head_bwd(C.byref(a3), C.byref(a1), None) == -1 and b"dzmid" in lib.wmd_last_error()
    a1.dz = 1
    a3.head[0].nrows, a3.n_out = 1, 6
    assert lib.wmd_head_bwd(C.byref(a3), C.byref(a1), None) == -3                                  # the merged form: 3-channel heads only


# ---- NYUv2 Model(opts) (NYUv2/model.py:12-71) ------------------------------------------------------------------------------
def test_nyu_model_picks_encoder_and_decoder_like_the_reference():
    from types import SimpleNamespace as NS
    import torch
    from wavelet_monodepth_amd import _lib, nyu

    def opts(**kw):
        base = dict(encoder_type="resnet", num_layers=18, pretrained_encoder=False, normalize_input=True, use_wavelets=True,
                    use_sparse=False, use_224=False, dw_waveconv=False, dw_upconv=False)
        base.update(kw)
        return NS(**base)

    def depthwise(dec):
        return any(".conv.0.0.weight" in k or k.startswith("conv.0.0") or ".0.0.weight" in k for k in dec.state_dict())

    cases = [(dict(), nyu.DecoderWave, False), (dict(use_224=True), nyu.DecoderWave224, False),
             (dict(use_sparse=True), nyu.SparseDecoderWave, False), (dict(use_wavelets=False), nyu.Decoder, False),
             (dict(use_wavelets=False, use_224=True), nyu.Decoder224, False),
             (dict(use_wavelets=False, dw_upconv=True), nyu.Decoder, True),
             (dict(dw_waveconv=True, dw_upconv=True), nyu.DecoderWave, True)]
    for kw, cls, dw in cases:
        m = nyu.Model(opts(**kw))
        assert type(m.decoder) is cls and m.use_sparse == bool(kw.get("use_sparse")), kw
        assert depthwise(m.decoder) == dw, kw
        assert list(m.encoder.num_ch_enc) == [64, 64, 128, 256, 512]
    assert list(nyu.Model(opts(num_layers=50)).encoder.num_ch_enc) == [64, 256, 512, 1024, 2048]
    # a namespace that predates --use_sparse (model.py:44-46) means dense; the caller's namespace is only read
    o = opts()
    del o.use_sparse
    m = nyu.Model(o)
    assert type(m.decoder) is nyu.DecoderWave and m.use_sparse is False and not hasattr(o, "use_sparse")
    with pytest.raises(NotImplementedError):
        nyu.Model(opts(use_sparse=True, use_224=True))                     # model.py:41-42
    with pytest.raises(NotImplementedError):
        nyu.Model(opts(encoder_type="vgg"))
    # config 5: DenseNet161 (densenet_encoder.py:13-33) and the two MobileNetV2 flavours (model.py:25-30)
    from wavelet_monodepth_amd import encoders
    m = nyu.Model(opts(encoder_type="densenet"))
    assert type(m.encoder) is encoders.DenseEncoder and list(m.encoder.num_ch_enc) == [96, 96, 192, 384, 2208]
    assert m.decoder.conv2.conv.weight.shape == (1104, 2208, 3, 3)
    assert sum(p.numel() for p in m.encoder.parameters()) == 28681000      # torchvision densenet161
    for kind, last in (("mobilenet", 1280), ("mobilenet_light", 160)):
        m = nyu.Model(opts(encoder_type=kind))
        assert list(m.encoder.num_ch_enc) == [32, 24, 32, 64, last]

    class Enc(torch.nn.Module):                                             # what the reference's DenseEncoder exposes
        num_ch_enc = [96, 96, 192, 384, 2208]

        def forward(self, x):
            return [torch.zeros(1, c, 64 >> (k + 1), 96 >> (k + 1)) for k, c in enumerate(self.num_ch_enc)]

    m = nyu.Model(opts(encoder_type="densenet", use_sparse=True), encoder=Enc())
    assert m.decoder.conv2.conv.weight.shape == (1104, 2208, 3, 3) and m.use_sparse
    with pytest.raises(_lib.WmdError):                                      # encoder on the host, decoder refuses: no CPU fallback
        m(torch.zeros(1, 3, 64, 96), 0.1)
    # the ResNet encoder itself is ordinary PyTorch and does NOT normalise (the reference's loop discards its result)
    enc = nyu.NyuResnetEncoder(18, normalize_input=True).eval()
    x = torch.rand(1, 3, 64, 96)
    with torch.no_grad():
        feats = enc(x)
        e = enc.encoder
        assert torch.equal(feats[0], e.relu(e.bn1(e.conv1(x))))
    assert [tuple(f.shape[1:]) for f in feats] == [(64, 32, 48), (64, 16, 24), (128, 8, 12), (256, 4, 6), (512, 2, 3)]


# ---- the KITTI trainer's parameter groups (trainer.py:74-75 + pyt_utils.py:12-28) ---------------------------------------
def test_convs_attribute_satisfies_the_trainers_parameter_group_rule():
    """The trainer walks `depth.convs.items()` and splits every value's parameters into a weight-decay group (Conv2d /
    Linear weights) and a no-decay group (their biases), ASSERTING that no parameter is left over.  Restated here: every
    parameter of every `.convs` value must be the weight or bias of an nn.Conv2d / nn.Linear, and the `.convs` values must
    cover exactly the decoder's parameters (what is not in `.convs` would never be optimised)."""
    import torch.nn as nn
    from wavelet_monodepth_amd.kitti import DepthDecoder, DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder
    for cls in (DepthDecoder, DepthWaveProgressiveDecoder, SparseDepthWaveProgressiveDecoder):
        dec = cls(np.array(R18))
        seen = set()
        for key, module in dec.convs.items():
            decay, no_decay = [], []
            for m in module.modules():
                if isinstance(m, (nn.Linear, nn.Conv2d, nn.Conv3d)):
                    decay.append(m.weight)
                    if m.bias is not None:
                        no_decay.append(m.bias)
            assert len(list(module.parameters())) == len(decay) + len(no_decay), key
            seen.update(id(p) for p in decay + no_decay)
        assert seen == {id(p) for p in dec.parameters()}, cls.__name__
        # keys are the reference's tuples: ("upconv", i, j), ("waveconv", i, j) / ("dispconv", s)
        assert all(isinstance(k, tuple) and k[0] in ("upconv", "waveconv", "dispconv") for k in dec.convs)


def test_kitti_factories_follow_the_reference_signatures():
    """network_constructors.py:12-40: make_depth_encoder(opts), make_depth_decoder(encoder, opts)."""
    from types import SimpleNamespace as NS
    from wavelet_monodepth_amd import kitti
    opts = NS(encoder_type="resnet", num_layers=18, weights_init="scratch", use_wavelets=True, use_sparse=False, scales=[0, 1, 2, 3])
    enc = kitti.make_depth_encoder(opts)
    assert list(enc.num_ch_enc) == [64, 64, 128, 256, 512]
    assert type(kitti.make_depth_decoder(enc, opts)) is kitti.DepthWaveProgressiveDecoder
    opts.use_sparse = True
    assert type(kitti.make_depth_decoder(enc, opts)) is kitti.SparseDepthWaveProgressiveDecoder
    opts.use_wavelets = False
    base = kitti.make_depth_decoder(enc, opts)
    assert type(base) is kitti.DepthDecoder and list(base.scales) == [0, 1, 2, 3]
    mob = kitti.make_depth_encoder(NS(encoder_type="mobilenet", num_layers=18, weights_init="scratch"))
    assert list(mob.num_ch_enc) == [32, 24, 32, 64, 1280]                  # mobilenetv2_encoder.py:142
    assert list(kitti.make_depth_encoder(NS(encoder_type="mobilenet_light", num_layers=18, weights_init="scratch")).num_ch_enc)[-1] == 160
    with pytest.raises(NotImplementedError):
        kitti.make_depth_encoder(NS(encoder_type="vgg", num_layers=18, weights_init="scratch"))
    with pytest.raises(RuntimeError):                     # "pretrained" needs a download: refused loudly
        kitti.make_depth_encoder(NS(encoder_type="resnet", num_layers=18, weights_init="pretrained"))
    # explicit form used by tools/test_simple.py
    assert type(kitti.make_depth_decoder(enc.num_ch_enc, range(4), use_wavelets=True, use_sparse=True)) \
        is kitti.SparseDepthWaveProgressiveDecoder
    assert type(kitti.make_depth_decoder(enc.num_ch_enc, range(4))) is kitti.DepthDecoder


# ---- encoders stay PyTorch, with torchvision's state_dict names (released / ImageNet checkpoints load with strict=True) ------
def test_densenet161_and_mobilenetv2_state_dict_names_follow_torchvision():
    """torchvision is not installed here, so the manifest is restated from its published architecture: densenet161 =
    growth 48, blocks (6, 12, 36, 24), 96 stem features, bn_size 4 (`features.denseblockK.denselayerL.{norm1,conv1,norm2,
    conv2}`, `features.transitionK.{norm,conv}`, `features.norm5`, `classifier`); mobilenet_v2 `features.N...` as the
    reference's own MobileNetV2Encoder builds them (mobilenetv2_encoder.py:112-136)."""
    import torch
    from wavelet_monodepth_amd.encoders import DenseEncoder, MobileNetV2Encoder
    sd = DenseEncoder().state_dict()
    want = {"original_model.features.conv0.weight": (96, 3, 7, 7), "original_model.classifier.weight": (1000, 2208),
            "original_model.features.norm5.weight": (2208,)}
    c = 96
    for k, n in enumerate((6, 12, 36, 24), 1):
        for l in range(1, n + 1):
            p = "original_model.features.denseblock%d.denselayer%d." % (k, l)
            want[p + "norm1.weight"] = (c,)
            want[p + "conv1.weight"] = (192, c, 1, 1)
            want[p + "norm2.running_var"] = (192,)
            want[p + "conv2.weight"] = (48, 192, 3, 3)
            c += 48
        if k < 4:
            want["original_model.features.transition%d.conv.weight" % k] = (c // 2, c, 1, 1)
            want["original_model.features.transition%d.norm.bias" % k] = (c,)
            c //= 2
    for name, shape in want.items():
        assert name in sd and tuple(sd[name].shape) == shape, name
    # 4 tensors per conv-less BN (+ num_batches_tracked), no unexpected extras: every key belongs to a known family
    import re
    pat = re.compile(r"original_model\.(features\.(conv0|norm0|norm5|denseblock[1-4]\.denselayer\d+\.(norm[12]|conv[12])|"
                     r"transition[1-3]\.(norm|conv))|classifier)\.(weight|bias|running_mean|running_var|num_batches_tracked)$")
    assert all(pat.match(k) for k in sd), [k for k in sd if not pat.match(k)][:5]
    msd = MobileNetV2Encoder().state_dict()
    assert tuple(msd["features.0.0.weight"].shape) == (32, 3, 3, 3)
    assert tuple(msd["features.1.conv.0.0.weight"].shape) == (32, 1, 3, 3)          # t = 1 block: depthwise first
    assert tuple(msd["features.2.conv.0.0.weight"].shape) == (96, 16, 1, 1)         # expansion 6
    assert tuple(msd["features.17.0.weight"].shape) == (1280, 160, 1, 1)            # the reference drops the 320 stage
    x = torch.rand(1, 3, 64, 96)
    with torch.no_grad():
        feats = DenseEncoder().eval()(x)
    assert [tuple(f.shape[1:]) for f in feats] == [(96, 32, 48), (96, 16, 24), (192, 8, 12), (384, 4, 6), (2208, 2, 3)]


# ---- round 2 host logic ------------------------------------------------------------------------------------------------------
def test_lazy_ops_dict_resolves_on_first_access_only():
    """sparse_ops.LazyOpsDict: the reference's integer op counts are computed when somebody looks at them -- plain map reads
    never trigger the resolver, every dict-style access to a lazy entry does, exactly once."""
    from wavelet_monodepth_amd.sparse_ops import LazyOpsDict
    calls = []

    def resolver():
        calls.append(1)
        return {"total_ops": 42, ("total_ops", 0): 7}

    d = LazyOpsDict({("disp", 0): "map"})
    d.set_lazy(["total_ops", ("total_ops", 0)], resolver)
    assert len(d) == 3 and d[("disp", 0)] == "map" and ("total_ops" in d) and not calls   # membership / plain reads: no resolution
    assert d["total_ops"] == 42 and d[("total_ops", 0)] == 7 and len(calls) == 1
    assert dict(d.items())["total_ops"] == 42 and d.get("total_ops") == 42 and d.get("nope", 5) == 5 and len(calls) == 1
    e = LazyOpsDict({("disp", 0): "map"})
    e.set_lazy(["total_ops", ("total_ops", 0)], resolver)
    assert e == d and len(calls) == 2                                              # comparison resolves the other side once
    assert sorted(map(str, e.values())) == sorted(map(str, d.values())) and e.copy() == dict(d)


def test_lazy_ops_dict_copies_never_expose_the_placeholders():
    """Round-2 ADVICE: CPython copies a dict subclass through its raw storage unless __iter__ is overridden, so `dict(out)`,
    `{**out}`, `out | x`, pop / setdefault and pickle used to hand out the None placeholders of the lazy entries."""
    import copy
    import pickle
    from wavelet_monodepth_amd.sparse_ops import LazyOpsDict

    def fresh():
        d = LazyOpsDict({("disp", 0): "map"})
        d.set_lazy(["total_ops", ("total_ops", 0)], lambda: {"total_ops": 42, ("total_ops", 0): 7})
        return d
    want = {("disp", 0): "map", "total_ops": 42, ("total_ops", 0): 7}
    assert dict(fresh()) == want and {**fresh()} == want and (fresh() | {}) == want and ({} | fresh()) == want
    u = {}
    u.update(fresh())
    assert u == want
    assert set(fresh()) == set(want) and list(fresh().keys()) == list(want)
    assert fresh().pop("total_ops") == 42 and fresh().setdefault(("total_ops", 0), -1) == 7
    assert pickle.loads(pickle.dumps(fresh())) == want and copy.copy(fresh()) == want and copy.deepcopy(fresh()) == want
    assert "None" not in repr(fresh())


def test_bucket_groups_cover_every_parameter_once_in_backward_order():
    """ddp.bucket_groups: decoder first, then the encoder in reverse registration order, cut into <= bucket_bytes messages;
    works for ResNet, DenseNet161 (config 5: 233 MB of gradients) and MobileNetV2 alike."""
    from wavelet_monodepth_amd.ddp import bucket_groups
    from wavelet_monodepth_amd.encoders import DenseEncoder, MobileNetV2Encoder, ResnetEncoder
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder
    from wavelet_monodepth_amd.nyu import DecoderWave
    for enc, dec in ((ResnetEncoder(50), None), (DenseEncoder(), "nyu"), (MobileNetV2Encoder(), None)):
        dec = DecoderWave(enc_features=enc.num_ch_enc) if dec == "nyu" else DepthWaveProgressiveDecoder(enc.num_ch_enc)
        groups = bucket_groups(enc, dec, bucket_bytes=25 << 20)
        flat = [p for _n, ps in groups for p in ps]
        want = list(enc.parameters()) + list(dec.parameters())
        assert len(flat) == len(want) and {id(p) for p in flat} == {id(p) for p in want}
        assert groups[0][0] == "decoder" and all(n.startswith("decoder") or n.startswith("encoder") for n, _ in groups)
        first_enc = next(i for i, (n, _) in enumerate(groups) if n.startswith("encoder"))
        assert all(n.startswith("decoder") for n, _ in groups[:first_enc]) and all(n.startswith("encoder") for n, _ in groups[first_enc:])
        # reverse registration order: the encoder's LAST parameter leads its first bucket, its stem weight closes the last one
        assert groups[first_enc][1][0] is list(enc.parameters())[-1] and groups[-1][1][-1] is list(enc.parameters())[0]
        for n, ps in groups:
            size = 4 * sum(p.numel() for p in ps)
            assert size <= (25 << 20) or len(ps) == 1, (n, size)
    dense_bytes = 4 * sum(p.numel() for p in DenseEncoder().parameters())
    assert 110e6 < dense_bytes < 120e6      # + 127 MB of decoder: the 233 MB exchange of config 5


def test_tuner_choose_picks_the_fastest_valid_candidate_and_caches_its_label(monkeypatch):
    """tuner.choose (the weight-gradient family's autotuner): invalid candidates are skipped, the winner's LABEL is cached,
    a cached label wins without timing, tuning off returns the library's own choice."""
    import torch
    from wavelet_monodepth_amd import tuner

    class FakeEvent:
        now = [0.0]

        def __init__(self, enable_timing=False):
            self.t = 0.0

        def record(self):
            self.t = FakeEvent.now[0]

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return other.t - self.t

    cost = {0: 5.0, -1: 3.0, 1: 1.0, 2: None, 3: 2.0, 9: 4.0}
    launched = []

    def launch(arg):
        launched.append(arg)
        if cost[arg] is None:
            return -3
        FakeEvent.now[0] += cost[arg]
        return 0

    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(tuner, "enabled", True)
    monkeypatch.setattr(tuner, "_cache_file", None)
    cands = [("library", 0), ("direct", -1), ("a", 1), ("b", 2), ("c", 3)]
    key = "wgrad|test|%d" % id(cands)
    assert tuner.choose(key, cands, launch) == 1 and tuner._cache[key][0] == "a"
    n = len(launched)
    assert tuner.choose(key, cands, launch) == 1 and len(launched) == n          # cached: nothing is timed again
    assert tuner.choose(key, [("library", 0), ("zz", 9)], launch) == 9            # cached label gone from the table: re-tunes
    monkeypatch.setattr(tuner, "enabled", False)
    assert tuner.choose("wgrad|other", cands, launch) == 0


def test_train_step_graph_refuses_an_optimizer_that_cannot_be_captured():
    import torch
    from wavelet_monodepth_amd.graphs import TrainStepGraph
    p = torch.nn.Parameter(torch.zeros(3))
    with pytest.raises(ValueError, match="capturable=True"):
        TrainStepGraph(lambda: p.sum(), torch.optim.Adam([p], lr=1e-3))


def test_mobilenetv2_encoder_starts_from_the_reference_initialisation_and_honours_custom_relu6():
    """Round-2 ADVICE: weights_init='scratch' must start where the reference does (mobilenetv2_encoder.py:136,160-173: conv
    weights ~ N(0, sqrt(2 / (k k Cout))), BatchNorm 1 / 0), and `use_custom_relu6` (the composed ReLU6, :18-30) is honoured."""
    import torch
    from wavelet_monodepth_amd.encoders import MobileNetV2Encoder
    torch.manual_seed(0)
    enc = MobileNetV2Encoder()
    conv = enc.features[-1][0]                     # 1x1 -> 1280: many weights, a tight estimate of the spread
    want = (2.0 / (1 * 1 * conv.out_channels)) ** 0.5
    assert abs(float(conv.weight.std()) / want - 1) < 0.02 and abs(float(conv.weight.mean())) < 0.01 * want
    bn = enc.features[-1][1]
    assert bool((bn.weight == 1).all()) and bool((bn.bias == 0).all())
    custom = MobileNetV2Encoder(use_custom_relu6=True)
    assert list(custom.state_dict()) == list(enc.state_dict())
    custom.load_state_dict(enc.state_dict())
    enc.eval(), custom.eval()
    x = torch.randn(1, 3, 64, 96)
    with torch.no_grad():
        for a, b in zip(enc(x), custom(x)):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(a.abs().max()))


def test_resnet_encoder_can_defer_its_last_relu():
    """Encoder edge: with defer_last_relu the last feature comes back as a DeferredActivation whose
    activation is the block's final ReLU -- applied by the decoders' first convolution on load; under autograd the encoder
    returns the ordinary activated tensor.  Plain PyTorch on the CPU."""
    import torch
    from wavelet_monodepth_amd.encoders import ResnetEncoder
    from wavelet_monodepth_amd.layers import DeferredActivation, split_edge
    torch.manual_seed(0)
    for layers in (18, 50):
        enc = ResnetEncoder(layers, defer_last_relu=True).eval()
        ref = ResnetEncoder(layers).eval()
        ref.load_state_dict(enc.state_dict())
        x = torch.rand(1, 3, 64, 96)
        with torch.no_grad():
            got, want = enc(x), ref(x)
        assert isinstance(got[-1], DeferredActivation) and got[-1].act == "leaky" and got[-1].slope == 0.0
        assert float(got[-1].tensor.min()) < 0.0                      # really a pre-activation
        assert torch.equal(got[-1].activate(), want[-1])
        for a, b in zip(got[:-1], want[:-1]):
            assert torch.equal(a, b)
        feats, edge = split_edge(got)
        assert edge is got[-1] and feats[-1] is edge.tensor and split_edge(want)[1] is None
        assert torch.is_tensor(enc(x)[-1])                             # autograd on: ordinary path


# ---- round 6: host-side pieces added for VERDICT r5 #4 / ADVICE r5 ---------------------------------------------------------
def _load_bench():
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(__file__))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_deadline_reports_and_leaves_when_the_status_exchange_hangs():
    """bench.py's Deadline: a rank whose peers never reach the set-up status exchange must not wait forever.  On
    expiry every rank writes its record to stderr; rank 0 still prints the headline line (train.error = the record) and leaves
    with 0, the others leave with 3; a cancelled deadline does nothing."""
    import io, json, time
    bench = _load_bench()
    codes, out = [], io.StringIO()
    bench._EMERGENCY["line"] = {"metric": "m", "value": 1.0, "train": None}
    d = bench.Deadline(0.05, lambda: {"rank": 0, "error": "WmdError: wmd_comm_init"}, 0, _exit=codes.append, _out=out)
    time.sleep(0.4)
    assert codes == [0]
    line = json.loads(out.getvalue())
    assert line["value"] == 1.0 and line["train"]["error"]["error"].startswith("WmdError") and line["train"]["error"]["deadline_s"] == 0.05
    codes.clear()
    d = bench.Deadline(0.05, lambda: {"rank": 1, "error": None}, 1, _exit=codes.append, _out=out)
    time.sleep(0.4)
    assert codes == [3]                                   # a non-zero rank prints no line
    codes.clear()
    d = bench.Deadline(0.2, lambda: {"rank": 0, "error": None}, 0, _exit=codes.append, _out=out)
    d.cancel()
    time.sleep(0.4)
    assert codes == []
    bench._EMERGENCY["line"] = None


def test_list_tile_query_agrees_with_the_planner_table():
    """wmd_conv_list_tile_supported (what sparse_decoder.tile_for asks) and plan_conv's work-list filter are ONE
    predicate now; the two shapes with LIST instantiations must be reported, others not."""
    from wavelet_monodepth_amd import _lib
    l = _lib.lib()
    assert l.wmd_conv_list_tile_supported(8, 16) == 1 and l.wmd_conv_list_tile_supported(16, 16) == 1
    for th, tw in ((8, 32), (4, 32), (6, 40), (16, 32), (12, 40), (0, 0), (8, 8)):
        assert l.wmd_conv_list_tile_supported(th, tw) == 0, (th, tw)


def test_bind_inputs_routes_are_decided_on_the_host():
    """decoder._bound (round 6): the route of a graph-mode forward is host logic -- buffers themselves / a recurring address set /
    copy -- and is decided before anything is launched.  Checked on stand-in tensors that record their copies (no GPU here)."""
    from wavelet_monodepth_amd.kitti import DepthWaveProgressiveDecoder

    class T:          # the handful of tensor attributes _bound looks at
        def __init__(self, addr):
            self.addr, self.shape, self.device, self.dtype, self.copies = addr, (2, 4), "d", "f", 0
        def data_ptr(self): return self.addr
        def is_contiguous(self): return True
        def copy_(self, src, non_blocking=False): self.copies += 1

    dec = DepthWaveProgressiveDecoder(np.array(R18))
    dec.static_inputs = [T(1), T(2)]
    dec._ptr_max, dec._ptr_seen, dec._ptr_keys = 1, {}, set()
    dec.static_route = {"buffers": 0, "pointer_replay": 0, "copy": 0}
    a, b = [T(10), T(20)], [T(30), T(40)]
    assert dec._bound(dec.static_inputs) == (dec.static_inputs, True)
    assert dec._bound(a) == (dec.static_inputs, True) and dec.static_inputs[0].copies == 1      # first sighting: copy
    assert dec._bound(a) == (a, False)                                                          # second: its own capture, in place
    assert dec._bound(b)[0] is dec.static_inputs and dec._bound(b)[0] is dec.static_inputs      # the cap (1) is reached: b keeps copying
    assert dec._bound(a) == (a, False)
    assert dec.static_route == {"buffers": 1, "pointer_replay": 2, "copy": 3}
    c = [T(10), T(20)]
    c[1].shape = (9, 9)
    assert dec._bound(c) == (c, True)                                                           # other shapes: the ordinary keyed path
