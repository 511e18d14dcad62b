"""SparseDepthWaveProgressiveDecoder on MI355X — API-compatible with the reference class
(/root/reference/KITTI/networks/decoders/depth_decoder.py:171-428): same constructor, parameters and
state_dict keys as the dense decoder (checkpoints are interchangeable), `forward(input_features,
thresh_ratio=0.05, sparse_scales=[0,1,2,3])`, same output keys incl. the five mask families and the
`total_ops` op model.  Inference only; batch 1 like the reference, or (extension) a batch of frames decoded together.

The five masks of the coarsest level (always all ones) are views of one cached constant shared by every call: treat the returned
masks as read-only.

Differences that are not observable in the outputs: activations stay dense and zero-initialised instead of
being compacted (see include/wmd.h); the forward never waits for the GPU -- the python-int `total_ops` entries are
resolved from the device-side pixel counts on first access (sparse_ops.LazyOpsDict), so independent frames can be
decoded concurrently on several streams; nothing is printed (the reference prints 'sparse: i').
`sparse_scales`: a level outside the list runs densely.  The reference only survives lists whose sparse levels are the
finest ones (its dense branch keeps reading the stale dense activation of the last dense level and crashes on the channel
mismatch); here a dense level below a sparse one simply convolves the zero-initialised sparse activations.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib, ops, sparse_ops as S
from ..wavelets import IDWT
from ..graphs import GraphCache
from .depth_decoder import _build_wave_convs
from ..layers import split_edge


def _conv_ops(cin, cout, npix, k):
    # dense layers: (1 + k*k*cin*npix) * cout — the "1 +" sits inside the pixel product (reference :386-398)
    return (1 + k * k * cin * npix) * cout


def _round64(n):
    return (int(n) + 63) // 64 * 64


def _sparse_conv_ops(cin, cout, nnz_out, mid=None):
    ops_ = 0
    if mid is not None:                       # fused leading 1x1 (layers.py:405)
        cin_mid, cmid, nnz_in = mid
        ops_ += nnz_in * cin_mid * cmid + nnz_in * cmid
    ops_ += cin * 9 * nnz_out                 # gathered elements (layers.py:462)
    ops_ += (1 + 9 * cin) * nnz_out * cout    # layers.py:469
    return ops_


def level_static_ops(i, h, w, sparse, trunk=None, heads=()):
    """Pixel-count independent part of the reference's op model for decoder level i working on an h x w coarse grid
    (depth_decoder.py:296-417): threshold + dilation arithmetic, the four mask2idxmap calls of a sparse level or the
    dense layers of a dense one, and the IDWT.  trunk = ((Cin0, Cout0), (Cin1, Cout1)), heads = [(Cin, Cmid, Cin3,
    Cout3), ...] are only read for a dense level."""
    H2, W2 = 2 * h, 2 * w
    n = 0
    if i != 4:
        n += 3 * h * w                                # max |yh| over three bands + compare (:308-309)
    n += 25 * h * w + 100 * h * w                     # MaxPool 5x5 on the coarse grid, 5x5 on the fine grid (:313-319)
    if sparse:
        n += 2 * h * w + 2 * H2 * W2                  # four mask2idxmap calls (layers.py:388)
    else:
        (ci0, co0), (ci1, co1) = trunk
        n += _conv_ops(ci0, co0, h * w, 3) + _conv_ops(ci1, co1, H2 * W2, 3)
        for cin, cmid, cin3, cout3 in heads:
            n += _conv_ops(cin, cmid, H2 * W2, 1) + _conv_ops(cin3, cout3, H2 * W2, 3)
    n += 4 * (2 * H2) * (2 * W2)                      # IDWT output pixels x 4 (:373,417)
    return n


def resolve_total_ops(static_ops, counters, counts):
    """The reference's per-scale and total op counts from the static parts, the layer widths of every sparse level
    (counters: (level, _, (Cin0, Cout0), (Cin1, Cout1), (Cin_1x1, Cmid), (Cin3, Cout3))) and that level's pixel
    counts (counts[level] = (upconv0, upconv1, wavelet) mask sums).  -> ({scale: ops}, total)."""
    per_scale, total = {}, 0
    for i in sorted(static_ops, reverse=True):
        n = static_ops[i]
        for (lvl, _nnz, (ci0, co0_), (ci1, co1_), (cim, cm), (ci3, co3)) in counters:
            if lvl != i:
                continue
            n0, n1, nw = counts[lvl]
            n += _sparse_conv_ops(ci0, co0_, n0) + _sparse_conv_ops(ci1, co1_, n1)
            n += 2 * _sparse_conv_ops(ci3, co3, nw, mid=(cim, cm, n1))
        per_scale[i - 1] = n
        total += n
    return per_scale, total


class SparseDepthWaveProgressiveDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.J = 1
        self.inverse_wt = IDWT(wave="haar", mode="zero")
        self.convs = _build_wave_convs(self.num_ch_enc, self.num_ch_dec, use_skips)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()
        self._graph_mode = False
        self._graphs = GraphCache()
        self._edge = None
        self._side = None          # stream of the activation-pool fill
        self._ones = {}            # all-ones mask constants of the coarsest level, per (device, B, h, w)
        self._states = {}          # sparse_ops.LevelState of the work-list form, per input signature

    # ------------------------------------------------------------------------------------------
    def _dense_coefficients(self, x, i, with_ll):
        heads = ([0] if with_ll else []) + [1, -1]
        mods = [self.convs[("waveconv", i, j)] for j in heads]
        mid = ops.conv1x1_stacked_nograd(x, [m[0].conv.weight for m in mods], [m[0].conv.bias for m in mods],
                                         act="leaky", slope=0.1)
        offs, o = {}, 0
        for j, m in zip(heads, mods):
            offs[j] = o
            o += m[0].conv.weight.shape[0]
        yl = None
        if with_ll:
            c3 = mods[0][2].conv
            yl = ops.head3x3_nograd(mid, c3.weight.shape[1], offs[0], c3.weight, c3.bias, pad="reflect", mode=1,
                                    scale=2.0 ** i)
        cp, cn = self.convs[("waveconv", i, 1)][2].conv, self.convs[("waveconv", i, -1)][2].conv
        yh = ops.head3x3_nograd(mid, cp.weight.shape[1], offs[1], cp.weight, cp.bias, offs[-1], cn.weight, cn.bias,
                                pad="reflect", mode=2, scale=2.0 ** (i - 1))
        return yl, yh.unsqueeze(1)

    def _block_sparse(self, B):
        """Form of the sparse levels.  Default: block-sparse execution of the dense Winograd kernels over the pixel TILES that
        hold active pixels + the dense fused head kernels with the wavelet mask in their epilogue (batch 1: 0.30 ms against
        0.33 ms, batch 12 at 10 % density: 0.75 ms against 1.18 ms).  WMD_SPARSE_TILES=0: gather-GEMMs over compacted pixel
        lists, the literal form of KITTI/layers.py:337-507."""
        return os.environ.get("WMD_SPARSE_TILES", "1") != "0"

    def enable_graph(self, on=True):
        """Capture the whole device-side chain (≈35 launches, all pixel counts stay on the device) into one hipGraph
        per (inputs, threshold, scales) and replay it; only the python-int op model is computed on the host after."""
        self._graph_mode = bool(on)
        self._graphs.clear()
        self._states.clear()
        self._ones.clear()
        return self

    @torch.no_grad()
    def forward(self, input_features, thresh_ratio=0.05, sparse_scales=[0, 1, 2, 3], _force_masks=None):
        # The reference asserts batch 1 ("works with single input only", depth_decoder.py:297).  Extension: a batch decodes
        # B frames -- each with its own coefficient range, masks, pixel lists and counts -- through the SAME launches (one
        # 640x192 frame is a few dozen to a few hundred 16-pixel tiles per launch and cannot fill 256 CUs); every per-frame
        # output equals the batch-1 result, and the `total_ops` entries become lists with one integer per frame.
        input_features, edge = split_edge(input_features)       # encoder edge: the last feature may be a pre-activation
        self._edge = edge
        S._on_gpu(*input_features)      # fails loudly on CPU tensors: there is no fallback path
        forced_on_device = _force_masks is None or all(m.is_cuda for m in _force_masks.values())
        graph = self._graph_mode and forced_on_device
        state = self._level_state(input_features, sparse_scales, _force_masks, graph)
        if state is not None:
            state.before_forward()
        if graph:
            thr, scales = float(thresh_ratio), tuple(sparse_scales)
            lv = sorted(_force_masks) if _force_masks else []      # injected masks are live inputs of the graph too
            nf = len(input_features)
            res = self._graphs.run(lambda f: self._device_chain(f[:nf], thr, scales, dict(zip(lv, f[nf:])) if lv else None, state),
                                   list(input_features) + [_force_masks[i] for i in lv], self.parameters(),
                                   extra_key=(thr, scales, tuple(lv), state is not None) + (("edge",) + edge.key() if edge is not None else ()))
            out, counters, static_ops = dict(res[0]), res[1], res[2]
            if state is not None:
                state.seq += 1          # the replay (the warm-up executions of a capture counted themselves)
        else:
            out, counters, static_ops = self._device_chain(input_features, thresh_ratio, sparse_scales, _force_masks, state)
        res = self._host_op_model(out, counters, static_ops, state)
        self.outputs = res      # depth_decoder.py:293,379,423,427: the op counts live in self.outputs too
        return res

    # ------------------------------------------------------------------------------------------
    def _lists_ok(self, sparse_scales, _force_masks):
        """The work-list form (default) serves the tile form when the sparse levels are the finest ones (a dense level BELOW a
        sparse one would read the never-refilled activation pool), every sparse level's heads run on the fused kernels and
        level 4 is not injected.  WMD_SPARSE_LISTS=0: the round-3 tile form (pool fill + per-block mask tests + count copy)."""
        if os.environ.get("WMD_SPARSE_LISTS", "1") == "0" or not self._block_sparse(0) or not self.use_skips:
            return False
        if _force_masks is not None and 4 in _force_masks:
            return False
        if any(i > 3 for i in set(sparse_scales)):
            # level 4 has no low-pass plane to threshold: the reference and the tile form fail an assertion there -- so must this
            # form (it used to drop the level silently and run it densely); 0 is ignored by every form
            return False
        lv = sorted(i for i in set(sparse_scales) if 1 <= i <= 3)
        if not lv or lv != list(range(1, lv[-1] + 1)):
            return False
        widths = [int(self.convs[("upconv", i, 1)].conv.conv.weight.shape[0]) for i in lv]
        return all(wd in ops.FUSED_HEAD_WIDTHS and wd % 8 == 0 for wd in widths) and int(self.num_ch_dec[4]) in ops.FUSED_HEAD_WIDTHS

    def _level_state(self, input_features, sparse_scales, _force_masks, graph):
        """The persistent device state of the work-list form for this input signature (None: another form runs).  In graph
        mode one state per captured graph (keyed like the graph: the captured launches hold its pointers; kept alive by the
        graph entry), else one per (stream, shapes)."""
        if not self._lists_ok(sparse_scales, _force_masks):
            return None
        x = input_features[-1]
        B, dev = x.shape[0], x.device
        lv = sorted(i for i in set(sparse_scales) if 1 <= i <= 3)
        shapes = tuple(tuple(f.shape) for f in input_features)
        if graph:
            key = ("g", dev.index, tuple(f.data_ptr() for f in input_features), shapes, tuple(lv))
        else:
            key = ("e", dev.index, torch.cuda.current_stream(dev).cuda_stream, shapes, tuple(lv))
        st = self._states.get(key)
        if st is None:
            if len(self._states) >= 64:
                self._states.clear()
                self._graphs.clear()       # captured launches hold the states' pointers
            pool_floats = 0
            for i in lv:
                hh, ww = input_features[i].shape[-2:]
                c0, c1 = (self.convs[("upconv", i, j)].conv.conv.weight.shape[0] for j in (0, 1))
                pool_floats += _round64(B * c0 * hh * ww) + _round64(B * c1 * 4 * hh * ww)
            st = S.LevelState(dev, B, len(lv), pool_floats)
            self._states[key] = st
        return st

    def _device_chain(self, input_features, thresh_ratio, sparse_scales, _force_masks, state=None):
        if state is not None:
            return self._device_chain_lists(input_features, thresh_ratio, sparse_scales, _force_masks, state)
        out = {}
        S._on_gpu(*input_features)      # fails loudly on CPU tensors: there is no fallback path
        x = input_features[-1].contiguous()
        dev = x.device
        B = x.shape[0]           # frames decoded together: every frame has its own range, masks, pixel lists and counts
        sparse_scales = list(sparse_scales)
        counters = []          # (level, nnz tensor [B,3]) resolved into python ints lazily
        static_ops = {}
        yl = yh = None
        xbuf = None            # dense [B,C,h,w] activations carried between levels
        # every zero-initialised activation plane of the sparse levels comes out of ONE pooled buffer (one fill per
        # forward instead of four per level; at batch 1 the chain is launch-latency bound)
        pool_floats = 0
        tiles = self._block_sparse(B)
        fused_heads = {}       # level -> the level's heads run on the dense fused head kernels (tile mode, supported width)
        for i in sparse_scales:
            if 1 <= i <= 3:
                hh, ww = input_features[i].shape[-2:]
                c0w, c1w = self.convs[("upconv", i, 0)].conv.conv.weight, self.convs[("upconv", i, 1)].conv.conv.weight
                cm = self.convs[("waveconv", i, 1)][0].conv.weight.shape[0]
                fused_heads[i] = tiles and int(c1w.shape[0]) in ops.FUSED_HEAD_WIDTHS
                pool_floats += sum(_round64(B * c * n) for c, n in ((c0w.shape[0], hh * ww), (c1w.shape[0], 4 * hh * ww)) +
                                   (() if fused_heads[i] else ((2 * cm, 4 * hh * ww), (3, 4 * hh * ww))))
        n_sparse = len(fused_heads)
        nnz_off = pool_floats
        pool_floats += _round64(n_sparse * B * 3)       # the pixel counts of every sparse level share the fill (int32 zeros)
        # ... filled on a side stream while the dense coarsest level runs when it is large (41 us at batch 12; at batch 1 the
        # fill takes 9 us and so does the stream join); joined before the first sparse level takes its first plane
        pool, pool_used, fill_stream = None, [0], [None]
        if n_sparse:
            pool = torch.empty(pool_floats, device=dev)
            cur = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side = torch.cuda.Stream(dev)
            if pool_floats * 4 >= (16 << 20):
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side):
                    pool.zero_()
                if not torch.cuda.is_current_stream_capturing():
                    pool.record_stream(self._side)
                fill_stream[0] = self._side
            else:                      # one frame: the join of a second stream costs as much as the fill itself
                pool.zero_()

        def join_fill():
            if fill_stream[0] is not None:
                torch.cuda.current_stream(dev).wait_stream(fill_stream[0])
                fill_stream[0] = None

        def zeros(*shape):
            n = int(np.prod(shape))
            v = pool[pool_used[0]:pool_used[0] + n].view(*shape)
            pool_used[0] += _round64(n)          # every plane set starts 256-byte aligned
            return v

        all_nnz = pool[nnz_off:nnz_off + n_sparse * B * 3].view(torch.int32).view(n_sparse, B, 3) if n_sparse else None

        SPECS = [(1, 1), (1, 2), (2, 2), (2, 1), (2, 0)]   # lowres, upconv0, upsample, upconv1, wavelet (:311-319)
        for i in range(4, -1, -1):
            scale_ops = 0
            fused_idwt = None       # (next low-pass, disparity) when the level's head kernels already ran the synthesis
            h, w = x.shape[-2:] if xbuf is None else xbuf.shape[-2:]
            forced = _force_masks is not None and i in _force_masks
            # a level whose heads run on the fused kernels needs no pixel lists: its three counts (upconv0, upconv1, wavelet:
            # specs 1, 3, 4) are accumulated by the mask launch itself
            counted = (all_nnz[len(counters)], [1, 3, 4]) if (i in sparse_scales and fused_heads.get(i)) else None
            if i in sparse_scales:
                join_fill()         # the counts and every activation plane of the sparse levels live in the pool
            if i == 4 and not forced:
                # all-ones mask: every dilation of it is all ones too (MaxPool2d pads with -inf) -- one constant, five views
                ones = self._ones.get((dev, B, h, w))
                if ones is None:
                    ones = self._ones[(dev, B, h, w)] = torch.ones(B * (2 * h * w + 3 * 4 * h * w), device=dev, dtype=torch.uint8)
                lowres, upconv0 = ones[:B * h * w].view(B, h, w), ones[B * h * w:2 * B * h * w].view(B, h, w)
                o4 = 2 * B * h * w
                upsample_m, upconv1, wavelet = (ones[o4 + k * 4 * B * h * w:o4 + (k + 1) * 4 * B * h * w].view(B, 2 * h, 2 * w)
                                                for k in range(3))
            elif forced:
                mask = _force_masks[i].to(dev).reshape(-1, h, w).to(torch.uint8)
                if mask.shape[0] != B:           # one injected mask shared by every frame
                    mask = mask.expand(B, h, w)
                lowres, upconv0, upsample_m, upconv1, wavelet = S.dilate_multi(mask.contiguous(), SPECS, counts=counted)
            elif os.environ.get("WMD_SPARSE_UNFUSED_MASKS", "0") == "1" and B == 1:   # the three-launch form (parity tests)
                mask = S.mask_threshold(yh, S.minmax(yl), thresh_ratio)
                lowres, upconv0, upsample_m, upconv1, wavelet = [m.unsqueeze(0) for m in S.dilate_multi(mask, SPECS)]
                counted = None
            else:
                lowres, upconv0, upsample_m, upconv1, wavelet = [m.reshape(B, *m.shape[-2:]) for m in
                                                                 S.mask_level(yl, yh, thresh_ratio, SPECS, counts=counted)]
            H2, W2 = 2 * h, 2 * w
            b = lambda m: m.view(torch.bool).reshape(B, 1, *m.shape[-2:])
            out[("lowres_mask", i - 1)] = b(lowres)
            out[("upconv0_mask", i - 1)] = b(upconv0)
            out[("upsample_mask", i - 1)] = b(upsample_m)
            out[("upconv1_mask", i - 1)] = b(upconv1)
            out[("wavelet_mask", i - 1)] = b(wavelet)
            c0 = self.convs[("upconv", i, 0)].conv.conv if i > 0 else None
            c1 = self.convs[("upconv", i, 1)].conv.conv if i > 0 else None

            if i in sparse_scales:
                assert self.use_skips and i > 0 and yl is not None
                scale_ops = level_static_ops(i, h, w, True)
                nnz = all_nnz[len(counters)]
                if counted is None:     # pixel lists (gather-GEMM form) and / or counts by stream compaction
                    (co0, co1, cow), nnz = S.compact_multi([upconv0, upconv1, wavelet], nnz_out=nnz)   # coords [B,npix], counts [B,3]
                src = xbuf if xbuf is not None else x
                C0, C1_ = c0.weight.shape[0], c1.weight.shape[0]
                x0 = zeros(B, C0, h, w)
                skip = input_features[i - 1].contiguous()
                x1 = zeros(B, C1_, H2, W2)
                if tiles:
                    # several frames: the trunk convolutions run on the dense Winograd / MFMA kernels over the pixel TILES
                    # that contain active pixels (tile list built on the device), input support and output mask folded
                    # into the patch gather / the epilogue -- same values as the gather-GEMM, dense-kernel throughput
                    ops._conv_fwd_raw(src, None, ops.pack_weights(c0.weight), c0.bias, C0, 3, "reflect", "elu", 0.0, 1,
                                      ops.pack_weights_wino(c0.weight), in_mask=lowres, out_mask=upconv0, out=x0)
                    ops._conv_fwd_raw(x0, skip, ops.pack_weights(c1.weight), c1.bias, C1_, 3, "reflect", "elu", 0.0, 2,
                                      ops.pack_weights_wino(c1.weight), in_mask=upsample_m, out_mask=upconv1, out=x1,
                                      in_mask_2x2=True)   # MaxPool5(upsample(m)) = upsample(MaxPool3(m)): constant on 2x2 blocks
                else:
                    S.sparse_conv(x0, src, ops.pack_weights(c0.weight), c0.bias, C0, 3, co0, nnz.data_ptr(), h * w,
                                  in_mask=lowres, pad="reflect", act="elu", nnz_stride=3)
                    S.sparse_conv(x1, x0, ops.pack_weights(c1.weight), c1.bias, C1_, 3, co1, nnz.data_ptr() + 4, H2 * W2,
                                  x2=skip, up1=2, in_mask=upsample_m, pad="reflect", act="elu", nnz_stride=3)
                hp, hn = self.convs[("waveconv", i, 1)], self.convs[("waveconv", i, -1)]
                Cm = hp[0].conv.weight.shape[0]
                if tiles and C1_ in ops.FUSED_HEAD_WIDTHS:
                    # heads of the tile mode: the dense fused head kernels (chained 1x1 -> tap-partial 3x3 -> sigmoid -> Haar
                    # synthesis) with the wavelet mask applied to yh in their epilogue.  Same values as the gather form: every
                    # 3x3 neighbour of a wavelet-mask pixel (reflection included) lies inside upconv1 = its radius-1 dilation,
                    # where x1 holds the sparse activations, so the mid channels those taps read are the ones layers.py:405
                    # computes; mid values outside the support are never read by a surviving output.
                    unpack = lambda m: (m[0].conv.weight, m[0].conv.bias, m[2].conv.weight, m[2].conv.bias)
                    yh, yl_next, disp_i = ops.head_fused_level_nograd(x1, unpack(hp), unpack(hn), scale=2.0 ** (i - 1), yl=yl,
                                                                      disp_scale=1.0 / 2 ** (i - 1), clamp01=True,
                                                                      yh_mask=wavelet.contiguous())
                    fused_idwt = (yl_next, disp_i)
                else:
                    # heads: stacked 1x1 + LeakyReLU on the upconv1 support, dual 3x3 + sigmoid on the wavelet mask
                    wstack, bstack = ops.stacked_pack([hp[0].conv.weight, hn[0].conv.weight], [hp[0].conv.bias, hn[0].conv.bias])
                    mid = zeros(B, 2 * Cm, H2, W2)
                    S.sparse_conv(mid, x1, wstack, bstack, 2 * Cm, 1, co1, nnz.data_ptr() + 4, H2 * W2, act="leaky", slope=0.1,
                                  nnz_stride=3)
                    yh_d = zeros(B, 3, H2, W2)
                    S.sparse_conv(yh_d, mid, ops.pack_weights(hp[2].conv.weight), hp[2].conv.bias, 3, 3, cow,
                                  nnz.data_ptr() + 8, H2 * W2, in_mask=upconv1, pad="reflect", act="sigmoid",
                                  out_scale=2.0 ** (i - 1), c1=Cm, c1_off=0, wp2=ops.pack_weights(hn[2].conv.weight),
                                  bias2=hn[2].conv.bias, c1_off2=Cm, nnz_stride=3)
                    yh = yh_d.unsqueeze(1)
                counters.append((i, nnz, (c0.weight.shape[1], C0), (c1.weight.shape[1], C1_),
                                 (hp[0].conv.weight.shape[1], Cm), (hp[2].conv.weight.shape[1], 3)))
                xbuf = x1
            else:
                src = xbuf if xbuf is not None else x
                if xbuf is None and getattr(self, "_edge", None) is not None:     # level 4 on the encoder's pre-activation
                    xd = ops.conv2d_pre_activated(src, self._edge.pre(), c0.weight, c0.bias, pad="reflect", act="elu")
                else:
                    xd = ops.conv2d_fused(src, c0.weight, c0.bias, pad="reflect", act="elu")
                skip = input_features[i - 1] if (self.use_skips and i > 0) else None
                cin1 = xd.shape[1] + (0 if skip is None else skip.shape[1])
                ux = ops.conv2d_fused(xd, c1.weight, c1.bias, x2=skip, up1=2, pad="reflect", act="elu")
                hds = [self.convs[("waveconv", i, j)] for j in ([0] if i == 4 else []) + [-1, 1]]
                scale_ops = level_static_ops(
                    i, h, w, False, ((src.shape[1], c0.weight.shape[0]), (cin1, c1.weight.shape[0])),
                    [(hd[0].conv.weight.shape[1], hd[0].conv.weight.shape[0], hd[2].conv.weight.shape[1],
                      hd[2].conv.weight.shape[0]) for hd in hds])
                if i == 4 and not forced and int(ux.shape[1]) in ops.FUSED_HEAD_WIDTHS:
                    # the dense level of every decode: the inference form of the heads (chained 1x1 -> tap-partial GEMMs, then
                    # one pass of 9-tap gather + sigmoid + combine + Haar synthesis + clamp: 3 launches instead of 5; the
                    # all-ones wavelet mask of this level needs no multiply)
                    unpack = lambda m: (m[0].conv.weight, m[0].conv.bias, m[2].conv.weight, m[2].conv.bias)
                    yh, yl_next, disp4, yl = ops.head_fused_level_nograd(
                        ux, unpack(self.convs[("waveconv", 4, 1)]), unpack(self.convs[("waveconv", 4, -1)]), scale=2.0 ** 3,
                        disp_scale=1.0 / 2 ** 3, clamp01=True, head_ll=unpack(self.convs[("waveconv", 4, 0)]), scale_ll=2.0 ** 4)
                    fused_idwt = (yl_next, disp4)
                else:
                    ll_new, yh = self._dense_coefficients(ux, i, with_ll=(i == 4))
                    if i == 4:
                        yl = ll_new
                    else:
                        yh = yh * wavelet.reshape(B, 1, 1, H2, W2)   # reference :272 (all ones at i == 4)
                xbuf = ux

            out[("wavelets", i - 1, "LL")] = yl
            out[("wavelets", i - 1, "LH")] = yh[:, :, 0]
            out[("wavelets", i - 1, "HL")] = yh[:, :, 1]
            out[("wavelets", i - 1, "HH")] = yh[:, :, 2]
            if fused_idwt is not None:
                yl, disp = fused_idwt
            else:
                yl, disp = ops.idwt_haar(yl, yh, disp_scale=1.0 / 2 ** (i - 1), clamp01=True)
            out[("disp", i - 1)] = disp
            static_ops[i] = scale_ops
            if i == 1:
                break
        return out, counters, static_ops

    def _device_chain_lists(self, input_features, thresh_ratio, sparse_scales, _force_masks, state):
        """The work-list form of the sparse levels (round 4).  Per level: ONE mask launch (threshold from the range keys the
        previous level's head epilogue left behind, the five masks, the active-tile lists of both trunk convolutions, the
        pixel counts into the ring) -> upconv(i,0) and upconv(i,1) over the listed tiles only, K split chosen on the device
        -> the fused head kernels, skipping pixel runs / wavefronts without an active pixel.  No pool fill, no count copy."""
        out = {}
        S._on_gpu(*input_features)
        x = input_features[-1].contiguous()
        dev, B = x.device, x.shape[0]
        lv = sorted((i for i in set(sparse_scales) if 1 <= i <= 3), reverse=True)
        capturing = torch.cuda.is_current_stream_capturing()
        pool_used = [0]

        def plane(*shape):
            n = int(np.prod(shape))
            v = state.pool[pool_used[0]:pool_used[0] + n].view(*shape)
            pool_used[0] += _round64(n)
            return v

        # Tile shape of a work-list launch: 8x16 (128-pixel workgroups: the finest granularity, most parallelism for one frame)
        # until the launch could hold more than WMD_SPARSE_BIG_FROM (item = tile x 32-channel slab) of them, 16x16 beyond (a batch
        # of frames: at full density the 8x16 form costs 35 % more than 256-pixel workgroups, while on contour masks both skip
        # about the same share of the work); WMD_SPARSE_TILE=HxW forces one shape.
        forced_tile = os.environ.get("WMD_SPARSE_TILE")
        big_from = int(os.environ.get("WMD_SPARSE_BIG_FROM", "700"))

        # ... and no list at all below WMD_SPARSE_LIST_FROM items (one frame; the coarse launches of a batch): the launch then runs the masked kernel the autotuner
        # picked for the shape (a block per tile tests its own mask bytes; host-chosen split-K) -- with a few dozen tiles on 256 CUs
        # nothing is gained by compacting them, and the tuned 16x16x4 / 32x32x2 kernels of round 3 are 10-20 % faster per launch
        # than the list kernel's one tile shape (r04_sparse_timelines.txt: 99 + 25 us of convolutions + reduces against 107 + 37)
        # (threshold swept at 12 frames, contour masks 0.10 / 0.03 / 0.01: 256 -> 0.633 ms, 500 -> 0.605, 800 -> 0.585, 1500 -> 0.630,
        #  no lists -> 0.689; dense 0.644-0.666: the lists pay on the two large launches of a batch -- upconv(2,1), upconv(1,1) --
        #  where tail rounds of whole tiles are what skipping loses, and cost on the small ones)
        list_from = int(os.environ.get("WMD_SPARSE_LIST_FROM", "800"))

        def tile_for(hh, ww, cout, cin=8, c_up=8):
            # a LIST launch needs the flattened staging (wino32_pure: every 8-channel chunk inside one source tensor) and a tile
            # shape the library has a work-list kernel for; anything else takes the mask form of the same level
            if cin % 8 or c_up % 8:
                return None
            if forced_tile:
                th, tw = (int(v) for v in forced_tile.split("x"))
                return (th, tw) if _lib.lib().wmd_conv_list_tile_supported(th, tw) else None
            items = B * (-(-hh // 8)) * (-(-ww // 16)) * (-(-cout // 32))
            if items < list_from:
                return None
            return (16, 16) if items > big_from else (8, 16)
        unpack = lambda m: (m[0].conv.weight, m[0].conv.bias, m[2].conv.weight, m[2].conv.bias)
        counters, static_ops = [], {}
        yl = yh = None
        xbuf = None
        keys_armed = False      # the head of the level above folded its LL range into state.keys
        prev_upconv1 = None     # upconv1 mask of the sparse level above (= the support of the activations this level reads)
        for i in range(4, 0, -1):
            h, w = x.shape[-2:] if xbuf is None else xbuf.shape[-2:]
            H2, W2 = 2 * h, 2 * w
            c0, c1 = self.convs[("upconv", i, 0)].conv.conv, self.convs[("upconv", i, 1)].conv.conv
            forced = _force_masks is not None and i in _force_masks
            next_sparse = (i - 1) in lv
            want_keys = next_sparse and not (_force_masks is not None and (i - 1) in _force_masks)
            if i in lv:
                k = lv.index(i)
                C0, C1_ = c0.weight.shape[0], c1.weight.shape[0]
                skip_c = input_features[i - 1].shape[1]
                specs = [(1, 1, 0, None), (1, 2, 1, tile_for(h, w, C0, c0.weight.shape[1])), (2, 2, 0, None),
                         (2, 1, 2, tile_for(H2, W2, C1_, c1.weight.shape[1], C0 if skip_c else 8)), (2, 0, 3, None)]
                if prev_upconv1 is not None:     # input support of upconv(i,0): lowres AND the previous sparse level's support
                    specs.append((1, 1, 0, None, prev_upconv1))
                if forced:
                    m0 = _force_masks[i].to(dev).reshape(-1, h, w).to(torch.uint8)
                    if m0.shape[0] != B:
                        m0 = m0.expand(B, h, w)
                    masks, lists = S.mask_level_lists(state, B, h, w, specs, mask0=m0, use_keys=keys_armed, counts_off=k * B * 3,
                                                      advance=(i == lv[-1]))
                else:
                    masks, lists = S.mask_level_lists(state, B, h, w, specs, thresh_ratio, yl=yl, yh=yh, use_keys=keys_armed,
                                                      counts_off=k * B * 3, advance=(i == lv[-1]))
                lowres, upconv0, upsample_m, upconv1, wavelet = masks[:5]
                lowres_in = masks[5] if prev_upconv1 is not None else lowres
                src = xbuf if xbuf is not None else x
                x0 = plane(B, C0, h, w)
                x1 = plane(B, C1_, H2, W2)
                skip = input_features[i - 1].contiguous()
                ops._conv_fwd_raw(src, None, ops.pack_weights(c0.weight), c0.bias, C0, 3, "reflect", "elu", 0.0, 1,
                                  ops.pack_weights_wino(c0.weight), in_mask=lowres_in, out_mask=upconv0, out=x0, out_tiles=lists[1])
                ops._conv_fwd_raw(x0, skip, ops.pack_weights(c1.weight), c1.bias, C1_, 3, "reflect", "elu", 0.0, 2,
                                  ops.pack_weights_wino(c1.weight), in_mask=upsample_m, out_mask=upconv1, out=x1,
                                  in_mask_2x2=True, out_tiles=lists[3])
                hp, hn = self.convs[("waveconv", i, 1)], self.convs[("waveconv", i, -1)]
                fold = want_keys and ops.head_level_folds_range_keys(C1_)
                yh, yl_next, disp_i = ops.head_fused_level_nograd(x1, unpack(hp), unpack(hn), scale=2.0 ** (i - 1), yl=yl,
                                                                  disp_scale=1.0 / 2 ** (i - 1), clamp01=True, yh_mask=wavelet,
                                                                  run_mask=upconv1, range_keys=state.keys if fold else None)
                keys_armed = fold
                prev_upconv1 = upconv1
                static_ops[i] = level_static_ops(i, h, w, True)
                counters.append((i, None, (c0.weight.shape[1], C0), (c1.weight.shape[1], C1_),
                                 (hp[0].conv.weight.shape[1], hp[0].conv.weight.shape[0]), (hp[2].conv.weight.shape[1], 3)))
                xbuf = x1
            else:
                # dense level (the coarsest one, and any level above the first sparse one)
                if i == 4:
                    ones = self._ones.get((dev, B, h, w))
                    if ones is None:
                        ones = self._ones[(dev, B, h, w)] = torch.ones(B * (2 * h * w + 3 * 4 * h * w), device=dev, dtype=torch.uint8)
                    lowres, upconv0 = ones[:B * h * w].view(B, h, w), ones[B * h * w:2 * B * h * w].view(B, h, w)
                    o4 = 2 * B * h * w
                    upsample_m, upconv1, wavelet = (ones[o4 + k * 4 * B * h * w:o4 + (k + 1) * 4 * B * h * w].view(B, H2, W2)
                                                    for k in range(3))
                else:
                    SPECS = [(1, 1), (1, 2), (2, 2), (2, 1), (2, 0)]
                    if forced:
                        m0 = _force_masks[i].to(dev).reshape(-1, h, w).to(torch.uint8)
                        if m0.shape[0] != B:
                            m0 = m0.expand(B, h, w)
                        lowres, upconv0, upsample_m, upconv1, wavelet = S.dilate_multi(m0.contiguous(), SPECS)
                    else:
                        lowres, upconv0, upsample_m, upconv1, wavelet = [m.reshape(B, *m.shape[-2:]) for m in
                                                                         S.mask_level(yl, yh, thresh_ratio, SPECS)]
                src = xbuf if xbuf is not None else x
                if xbuf is None and getattr(self, "_edge", None) is not None:
                    xd = ops.conv2d_pre_activated(src, self._edge.pre(), c0.weight, c0.bias, pad="reflect", act="elu")
                else:
                    xd = ops.conv2d_fused(src, c0.weight, c0.bias, pad="reflect", act="elu")
                skip = input_features[i - 1]
                ux = ops.conv2d_fused(xd, c1.weight, c1.bias, x2=skip, up1=2, pad="reflect", act="elu")
                hds = [self.convs[("waveconv", i, j)] for j in ([0] if i == 4 else []) + [-1, 1]]
                static_ops[i] = level_static_ops(
                    i, h, w, False, ((src.shape[1], c0.weight.shape[0]), (xd.shape[1] + skip.shape[1], c1.weight.shape[0])),
                    [(hd[0].conv.weight.shape[1], hd[0].conv.weight.shape[0], hd[2].conv.weight.shape[1],
                      hd[2].conv.weight.shape[0]) for hd in hds])
                fold = want_keys and ops.head_level_folds_range_keys(ux.shape[1])
                res = ops.head_fused_level_nograd(
                    ux, unpack(self.convs[("waveconv", i, 1)]), unpack(self.convs[("waveconv", i, -1)]), scale=2.0 ** (i - 1),
                    yl=None if i == 4 else yl, disp_scale=1.0 / 2 ** (i - 1), clamp01=True,
                    head_ll=unpack(self.convs[("waveconv", 4, 0)]) if i == 4 else None, scale_ll=2.0 ** 4,
                    yh_mask=None if i == 4 else wavelet.contiguous(), range_keys=state.keys if fold else None)
                keys_armed = fold
                if i == 4:
                    yh, yl_next, disp_i, yl = res
                else:
                    yh, yl_next, disp_i = res
                xbuf = ux
            b = lambda m: m.view(torch.bool).reshape(B, 1, *m.shape[-2:])
            out[("lowres_mask", i - 1)] = b(lowres)
            out[("upconv0_mask", i - 1)] = b(upconv0)
            out[("upsample_mask", i - 1)] = b(upsample_m)
            out[("upconv1_mask", i - 1)] = b(upconv1)
            out[("wavelet_mask", i - 1)] = b(wavelet)
            out[("wavelets", i - 1, "LL")] = yl
            out[("wavelets", i - 1, "LH")] = yh[:, :, 0]
            out[("wavelets", i - 1, "HL")] = yh[:, :, 1]
            out[("wavelets", i - 1, "HH")] = yh[:, :, 2]
            out[("disp", i - 1)] = disp_i
            yl = yl_next
        if not capturing:
            state.seq += 1      # an eager execution advanced the device's forward counter (a capture executes nothing)
        return out, counters, static_ops

    @staticmethod
    def _host_op_model(out, counters, static_ops, state=None):
        # ---- the reference's op model needs the pixel counts as python ints.  They are copied to pinned host memory
        # asynchronously and turned into the `total_ops` entries on first access: the forward itself never waits for the GPU
        out = S.LazyOpsDict(out)
        if state is not None:      # work-list form: the counts wait in the ring slot of this forward (number state.seq - 1)
            fetch = state.counts_fetcher(state.seq - 1)
        else:
            fetch = S.counts_to_host([nnz for (_lvl, nnz, *_rest) in counters])
        levels = [lvl for (lvl, *_rest) in counters]

        def resolver():
            flat = fetch()                                      # per level: B * 3 counts, frame-major
            B = len(flat[0]) // 3 if flat else 1
            frames = []
            for f in range(B):
                resolved = {lvl: cnt[3 * f:3 * f + 3] for lvl, cnt in zip(levels, flat)}
                frames.append(resolve_total_ops(static_ops, counters, resolved))
            if len(frames) == 1:
                per_scale, total = frames[0]
                res = {("total_ops", sc): n for sc, n in per_scale.items()}
                res["total_ops"] = total
            else:                                               # batched decode: one integer per frame
                res = {("total_ops", sc): [fr[0][sc] for fr in frames] for sc in frames[0][0]}
                res["total_ops"] = [fr[1] for fr in frames]
            return res

        out.set_lazy([("total_ops", i - 1) for i in static_ops] + ["total_ops"], resolver)
        return out
