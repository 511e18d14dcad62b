"""KITTI depth decoders on MI355X, API-compatible with the reference
(/root/reference/KITTI/networks/decoders/depth_decoder.py):

  DepthDecoder                       :18-69   sigmoid-disparity baseline
  DepthWaveProgressiveDecoder        :72-168  dense wavelet decoder
  SparseDepthWaveProgressiveDecoder  :171-428 threshold-gated sparse inference (see sparse_decoder.py)

Same constructor signatures, same `convs` OrderedDict keys / `decoder.N...` state_dict names, same
output dictionary keys.  All arithmetic runs in libwmd_hip.so; a CPU tensor raises.
"""
from collections import OrderedDict

import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..layers import Conv1x1, Conv3x3, ConvBlock, gated_backward_allowed, split_edge
from ..wavelets import IDWT
from ..graphs import GraphCache


def _wave_head(num_in, num_mid, num_out):
    # nn.Sequential(Conv1x1, LeakyReLU(0.1), Conv3x3(refl)) — indices 0 and 2 carry parameters
    return nn.Sequential(Conv1x1(num_in, num_mid), nn.LeakyReLU(0.1, inplace=True), Conv3x3(num_mid, num_out, use_refl=True))


def _build_wave_convs(num_ch_enc, num_ch_dec, use_skips):
    convs = OrderedDict()
    for i in range(4, 0, -1):
        num_ch_in = num_ch_enc[-1] if i == 4 else num_ch_dec[i + 1]
        convs[("upconv", i, 0)] = ConvBlock(num_ch_in, num_ch_dec[i], use_refl=True)
        num_ch_in = num_ch_dec[i]
        if use_skips and i > 0:
            num_ch_in += num_ch_enc[i - 1]
        convs[("upconv", i, 1)] = ConvBlock(num_ch_in, num_ch_dec[i], use_refl=True)
        if i == 4:
            convs[("waveconv", i, 0)] = _wave_head(num_ch_dec[i], num_ch_dec[i] // 4, 1)
        convs[("waveconv", i, 1)] = _wave_head(num_ch_dec[i], num_ch_dec[i], 3)
        convs[("waveconv", i, -1)] = _wave_head(num_ch_dec[i], num_ch_dec[i], 3)
    return convs


class DepthWaveProgressiveDecoder(nn.Module):
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
        self.tanh = nn.Tanh()
        self._graph_mode = False
        self._edge = None         # layers.DeferredActivation of the current call's last feature (encoder edge), if any
        self._graphs = GraphCache()
        self.stack_heads = os.environ.get("WMD_STACKED_HEADS", "1") == "1"   # training: one launch per stage over all heads of a level
        self.branch_trace = None   # set to a dict to record the LeakyReLU pieces of a training-mode forward
        self.fuse_heads = True   # inference: fused 1x1 -> 3x3 -> IDWT head kernels where the width allows (32/64/128)
        # opt-in: heads on a second stream -- 1: under graph capture only (a replayed graph serialises the fork: no gain), 2: eager
        # launches too (side stream of priority WMD_OVERLAP_PRIO; tools/probes/overlap_probe.py)
        self.overlap_heads = int(os.environ.get("WMD_OVERLAP_HEADS", "0"))
        self._side_stream = None
        # graph mode: one graph (default), or trunk / heads as graph segments on two streams (WMD_TWO_STREAM_GRAPHS=1).  The
        # two-stream replay won 2-3 % while the trunk ran on the 16x16x4 Winograd kernels; beside conv_wino32_kernel (two
        # 200-register blocks per CU) the side-stream heads no longer find idle CUs: 0.648 vs 0.625 ms (round 3, same box)
        self.two_stream_graphs = os.environ.get("WMD_TWO_STREAM_GRAPHS", "0") == "1"
        self.static_inputs = None      # bind_inputs(): decoder-owned input buffers of the replayed graphs
        self._ptr_max, self._ptr_seen, self._ptr_keys = 0, {}, set()     # ... and its recurring-address captures
        self.static_route = {"buffers": 0, "pointer_replay": 0, "copy": 0}
        self._segment_captures = 0
        self._segments = {}

    # -- pieces ------------------------------------------------------------------------------
    def _head_mid(self, x, key, x_gate=None, gated=True):
        # the 1x1's LeakyReLU output `mid` is consumed by the head's 3x3 only, whose backward returns d mid * leaky'(mid)
        # (x_gate on that side): the 1x1's backward takes its incoming gradient as dz -- unless a hook could hand `mid` to
        # somebody else (gated_backward_allowed)
        head = self.convs[key]
        return head[0](x, act="leaky", slope=0.1, x1_gate=x_gate, grad_is_dz=gated)

    def get_coefficients(self, input_features, scale=1, return_ll=False, _x_gate=None):
        """(LL, [LH, HL, HH]) from the features of level `scale` (reference :126-136).
        _x_gate (internal, set by forward()): input_features is the ELU output of this decoder's own trunk convolution."""
        if not torch.is_grad_enabled():
            return self._coefficients_stacked(input_features, scale, return_ll)
        stackable = all(self.convs[("waveconv", scale, j)][0].conv.weight.shape[0] % 16 == 0 for j in ([0] if return_ll else []) + [1])
        if self.stack_heads and stackable:   # (the stacked GEMM concatenates whole 16-channel tiles)
            hd = lambda j: (lambda m: (m[0].conv.weight, m[0].conv.bias, m[2].conv.weight, m[2].conv.bias))(self.convs[("waveconv", scale, j)])
            yh, yl, mid = ops.stacked_heads(input_features, hd(1), hd(-1), 2.0 ** (scale - 1), hd(0) if return_ll else None,
                                            2.0 ** scale, x_gate=_x_gate, return_mid=True)
            if self.branch_trace is not None:   # which LeakyReLU piece each element took (gradient-parity diagnostics)
                o = 0
                for j in ([0] if return_ll else []) + [1, -1]:
                    c = self.convs[("waveconv", scale, j)][0].conv.weight.shape[0]
                    self.branch_trace[("waveconv", scale, j)] = (mid[:, o:o + c] > 0).cpu()
                    o += c
            return yl, yh.unsqueeze(1)
        yl = None
        gated = gated_backward_allowed(self)
        leaky = ("leaky", 0.1) if gated else None
        if return_ll:
            mid = self._head_mid(input_features, ("waveconv", scale, 0), _x_gate, gated)
            c3 = self.convs[("waveconv", scale, 0)][2].conv
            yl = ops.head3x3(mid, c3.weight, c3.bias, pad="reflect", mode=1, scale=2.0 ** scale, x_gate=leaky)
        mp = self._head_mid(input_features, ("waveconv", scale, 1), _x_gate, gated)
        mn = self._head_mid(input_features, ("waveconv", scale, -1), _x_gate, gated)
        cp = self.convs[("waveconv", scale, 1)][2].conv
        cn = self.convs[("waveconv", scale, -1)][2].conv
        yh = ops.head3x3(mp, cp.weight, cp.bias, mn, cn.weight, cn.bias, pad="reflect", mode=2,
                         scale=2.0 ** (scale - 1), x_gate=leaky)
        return yl, yh.unsqueeze(1)

    def _coefficients_stacked(self, x, scale, return_ll):
        """Inference path: the 2-3 Conv1x1 of a level run as ONE stacked MFMA GEMM (x is read once) and the 3x3
        heads read their inputs as channel slices of its output."""
        order = ([0] if return_ll else []) + [1, -1]
        heads = [self.convs[("waveconv", scale, j)] for j in order]
        mid = ops.conv1x1_stacked_nograd(x, [h[0].conv.weight for h in heads], [h[0].conv.bias for h in heads],
                                         act="leaky", slope=0.1)
        offs, o = {}, 0
        for j, h in zip(order, heads):
            offs[j] = o
            o += h[0].conv.weight.shape[0]
        yl = None
        if return_ll:
            c3 = heads[0][2].conv
            yl = ops.head3x3_nograd(mid, c3.weight.shape[1], offs[0], c3.weight, c3.bias, pad="reflect", mode=1,
                                    scale=2.0 ** scale)
        cp = self.convs[("waveconv", scale, 1)][2].conv
        cn = self.convs[("waveconv", scale, -1)][2].conv
        yh = ops.head3x3_nograd(mid, cp.weight.shape[1], offs[1], cp.weight, cp.bias, offs[-1], cn.weight, cn.bias,
                                pad="reflect", mode=2, scale=2.0 ** (scale - 1))
        return yl, yh.unsqueeze(1)

    @property
    def capture_count(self):
        """Graph captures so far, in either graph mode (a caller that thrashes the replay cache sees it grow)."""
        return self._segment_captures + self._graphs.captures

    def bind_inputs(self, example_features, adopt=False, pointer_sets=4):
        """Static-input entry for graph replay.  The replay key of `enable_graph` is the identity of the input tensors, so a
        caller whose encoder returns fresh tensors every step (every real caller: trainer.py:240-241) would re-capture on
        every call.  After `bind_inputs`, the decoder has one static buffer per feature map (`decoder.input_buffers()`), and
        `forward` reaches the captured launches by the cheapest of three routes:

          1. handed the static buffers themselves -- an encoder whose last operators write into `decoder.input_buffers()`
             (`out=` / in-place), or `adopt=True`, which makes the caller's OWN tensors the static buffers -- it replays: no copy;
          2. handed tensors at device addresses it has seen before (a caching allocator in steady state returns the same blocks
             step after step), it captures once per recurring address set -- on the second sighting, at most `pointer_sets`
             sets, none retained -- and from then on replays that capture on the caller's tensors in place: no copy;
          3. otherwise it copies into the static buffers and replays the first capture (167.7 MB per config-2 batch: +0.07 ms).

        Inference only; a call with other shapes, or with autograd enabled, takes the ordinary path."""
        if adopt:
            for f in example_features:
                if not (f.is_cuda and f.is_contiguous() and f.dtype == torch.float32):
                    raise ValueError("bind_inputs(adopt=True) takes contiguous float32 device tensors (they become the graph's input buffers)")
            self.static_inputs = list(example_features)
        else:
            self.static_inputs = [torch.empty_like(f, memory_format=torch.contiguous_format) for f in example_features]
            for dst, src in zip(self.static_inputs, example_features):
                dst.copy_(src)
        self._ptr_max, self._ptr_seen, self._ptr_keys = int(pointer_sets), {}, set()
        self.static_route = {"buffers": 0, "pointer_replay": 0, "copy": 0}    # forwards by route (diagnostics; bench.py reports it)
        self._graph_mode = True
        self._graphs.shared_pool = torch.cuda.graph_pool_handle()   # the pointer-set captures share one memory pool
        with torch.no_grad():
            self.forward(self.static_inputs)          # warm-up + capture now, not inside the first timed call
        return self

    def input_buffers(self):
        """The static input buffers of `bind_inputs` (None before): whatever is written into them is what the next forward on
        them decodes -- the zero-copy hand-over for an encoder that can direct its outputs (`torch.add(a, b, out=buf)`, `buf.copy_`
        fused into its last kernel, ...)."""
        return self.static_inputs

    def _bound(self, input_features):
        """-> (features to run on, retain them in the graph cache?) for a graph-mode forward (see bind_inputs)."""
        st = self.static_inputs
        if st is None or len(st) != len(input_features):
            return input_features, True
        same = True
        for dst, src in zip(st, input_features):
            if src is dst:
                continue
            same = False
            if src.shape != dst.shape or src.device != dst.device or src.dtype != dst.dtype:
                return input_features, True
        if same:
            self.static_route["buffers"] += 1
            return st, True
        if self._ptr_max > 0 and all(f.is_contiguous() for f in input_features):
            pk = tuple(f.data_ptr() for f in input_features)
            if pk in self._ptr_keys:
                self.static_route["pointer_replay"] += 1
                return input_features, False
            n = self._ptr_seen.get(pk, 0) + 1
            if n >= 2 and len(self._ptr_keys) < self._ptr_max:      # a RECURRING address set: worth a capture of its own
                self._ptr_keys.add(pk)
                self._ptr_seen.pop(pk, None)
                self.static_route["pointer_replay"] += 1
                return input_features, False
            if len(self._ptr_seen) >= 64:
                self._ptr_seen.clear()
            self._ptr_seen[pk] = n
        for dst, src in zip(st, input_features):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        self.static_route["copy"] += 1
        return st, True

    def forward(self, input_features):
        # encoder edge: the last feature may arrive as a layers.DeferredActivation (pre-activation + what to apply on load)
        input_features, edge = split_edge(input_features)
        if edge is not None and torch.is_grad_enabled():
            input_features[-1], edge = edge.activate(), None        # training: the ordinary path on the activated tensor
        self._edge = edge
        if self._graph_mode and not torch.is_grad_enabled():
            input_features, retain = self._bound(input_features)
            if self.two_stream_graphs and edge is None and retain:
                self.outputs = self._forward_two_streams(input_features)
            else:
                self.outputs = self._graphs.run(self._forward_impl, input_features, self.parameters(),
                                                extra_key=("edge",) + edge.key() if edge is not None else (), retain_inputs=retain)
            return self.outputs
        return self._forward_impl(input_features)

    # -- two-stream replay ---------------------------------------------------------------------
    # A replayed hipGraph executes its nodes one after the other even when the capture forked onto a second stream
    # (tools/graph_concurrency_probe.py: two independent 32 us convolutions take 61 us forked inside one graph, 47 us as
    # two graphs replayed on two streams).  The forward has two dependency chains -- trunk T4 > T3 > T2 > T1 and heads
    # H4 > H3 > H2 > H1 with H_i after T_i -- and the coarse heads / coarse trunk layers are launches of 46-360 workgroups
    # on 256 CUs.  So the forward is captured as 8 graph segments: trunk segments (and H1, which has nothing left to
    # overlap with) replay on the caller's stream, H4..H2 on a side stream behind an event per level.  Segments of one
    # stream share a memory pool (they replay in capture order); the two streams use different pools, so a buffer freed
    # during one capture can never be handed to a segment that runs concurrently; tensors that cross streams stay alive.
    def _forward_two_streams(self, input_features):
        key = tuple((t.data_ptr(), tuple(t.shape)) for t in input_features) + tuple((p.data_ptr(), p._version) for p in self.parameters()) + (ops.pack_generation(),)
        ent = self._segments.get(key)
        if ent is None:
            if len(self._segments) >= 4:
                self._segments.clear()
            ent = self._capture_two_streams(list(input_features))
            self._segment_captures += 1
            self._segments[key] = ent
        trunk, heads, events, side, done, outputs, _keep = ent
        main = torch.cuda.current_stream()
        for i in (4, 3, 2):
            trunk[i].replay()
            events[i].record(main)
            side.wait_event(events[i])
            with torch.cuda.stream(side):
                heads[i].replay()
        done.record(side)
        trunk[1].replay()
        main.wait_event(done)
        heads[1].replay()
        return dict(outputs)

    def _capture_two_streams(self, feats):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):          # eager warm-up (autotuning, packed-weight caches)
            self._forward_impl(feats)
            self._forward_impl(feats)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        pool_main, pool_side = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        trunk, heads, keep = {}, {}, [feats]
        self.outputs = {}
        x, yl = feats[-1], None
        for i in range(4, 0, -1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool_main):
                x = self.convs[("upconv", i, 0)](x)
                skip = feats[i - 1] if self.use_skips else None
                x = self.convs[("upconv", i, 1)](x, skip=skip, up=2)
            trunk[i] = g
            keep.append(x)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool_main if i == 1 else pool_side):
                yl = self._level_heads(i, x, yl)
            heads[i] = g
            keep.append(yl)
        events = {i: torch.cuda.Event() for i in (4, 3, 2)}
        return trunk, heads, events, torch.cuda.Stream(), torch.cuda.Event(), dict(self.outputs), keep

    def enable_graph(self, on=True):
        """Inference only: capture the whole forward (≈25 kernel launches) into hipGraph(s) per input
        signature and replay it.  Outputs then live in static buffers that the next call overwrites."""
        self._graph_mode = bool(on)
        self._graphs.clear()
        self._segments.clear()
        if not on:
            self.static_inputs = None
            self._ptr_keys, self._ptr_seen = set(), {}
        return self

    def eager(self):
        """Context manager: forwards inside it are launched eagerly WITHOUT dropping the captured graphs (enable_graph(False)
        clears them).  For a profiling pass between capture and replay -- bench.py's per-kernel hipEvent pass -- that must not
        leave the GPU idle through a second capture right before the timed replays."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            was = self._graph_mode
            self._graph_mode = False
            try:
                yield self
            finally:
                self._graph_mode = was
        return _cm()

    def _forward_impl(self, input_features):
        self.outputs = {}
        ops.prepack_module(self)      # one launch for every weight image this pass (and its backward) will ask for
        x = input_features[-1]
        yl = None
        # Under hipGraph capture the wavelet heads run on a second stream: head(i) needs only x_i and the low-pass of
        # head(i+1), the trunk continues from x_i — two dependency chains.  The coarse-level heads are small launches
        # (46 - 720 workgroups) and so are the coarse trunk convolutions: side by side they fill each other's idle CUs.
        # (Capture only: there every buffer is static; the eager path stays on one stream.)  Measured on MI355X: 0.748 vs
        # 0.743 ms per step -- the replayed graph gains nothing from the fork, so this stays opt-in (WMD_OVERLAP_HEADS=1).
        overlap = bool(self.overlap_heads) and not torch.is_grad_enabled() and \
            (torch.cuda.is_current_stream_capturing() or self.overlap_heads >= 2)
        main = torch.cuda.current_stream() if overlap else None
        if overlap and self._side_stream is None:
            self._side_stream = torch.cuda.Stream(priority=int(os.environ.get("WMD_OVERLAP_PRIO", "0")))
        side = self._side_stream if overlap else None
        keep = []   # tensors that cross streams stay referenced until the streams have joined
        # training: every consumer of a trunk activation (the next trunk convolution, the heads' 1x1 convolutions) returns its
        # data gradient already multiplied by ELU'(activation), so no trunk convolution runs a separate activation-backward
        # pass (ops.conv2d_fused: x1_gate / grad_is_dz)
        elu = ("elu", 0.0) if (torch.is_grad_enabled() and gated_backward_allowed(self)) else None
        self._gated = elu is not None
        edge = getattr(self, "_edge", None)
        # dense inference: the heads of a level read only that level's trunk activation, so levels 4..2 are POSTPONED until
        # upconv(2,1) is done: ONE launch runs their chained first stages (round 6, ops.head_fused_gemm_multi_nograd: alone none of
        # these levels fills 256 CUs; config 2, batch 12: 0.082 -> 0.069 ms, one frame 0.202 -> 0.177 ms), ONE launch completes all
        # three (round 5, ops.head_shiftsum_chain_nograd: level k's synthesis output is level k+1's low-pass input, pixel for pixel).
        # Without the merged launch (WMD_HEAD_CHAIN_MULTI=0, odd plane sizes) every level launches its first stage in place and the
        # chained completion is used up to WMD_SHIFTSUM_CHAIN_MAX_PIXELS only (it then reads planes that have left the caches).
        widths = [int(self.num_ch_dec[k]) for k in (4, 3, 2)]
        plain = (not overlap) and (not torch.is_grad_enabled()) and self.fuse_heads and \
            bool(ops._lib.lib().wmd_head_level_supported(int(self.num_ch_dec[1])))
        chain_multi = plain and edge is None and ops.head_chain_multi_supported(widths) and \
            all((input_features[k].shape[2] * input_features[k].shape[3]) % 4 == 0 for k in (3, 2, 1))
        chain = plain and ops.shiftsum_chain_supported(
            widths, 0 if chain_multi else input_features[1].shape[0] * input_features[1].shape[2] * input_features[1].shape[3])
        pending, deferred = [], []
        # ... and (round 6) the completions of levels 4..2 run INSIDE level 1's launch: the streaming kernel's epilogue waves complete
        # the coarser levels over each unit's footprint first (ops.head_level_pyramid_nograd): one graph node and 21.7 us fewer
        f0 = input_features[0]
        pyramid = chain and ops.head_level_pyramid_supported(int(self.num_ch_dec[1]), f0.shape[0], f0.shape[2], f0.shape[3])
        for i in range(4, 0, -1):
            if i == 4 and edge is not None:
                x = self.convs[("upconv", 4, 0)](x, x1_pre=edge.pre())      # ReLU (+ affine) of the encoder's last block on load
            else:
                x = self.convs[("upconv", i, 0)](x, x1_gate=elu if i < 4 else None, grad_is_dz=elu is not None)
            skip = input_features[i - 1] if (self.use_skips and i > 0) else None
            x = self.convs[("upconv", i, 1)](x, skip=skip, up=2, x1_gate=elu, grad_is_dz=elu is not None)  # fused upsample + concat
            if chain and i >= 2:
                hd = lambda j: (lambda m: (m[0].conv.weight, m[0].conv.bias, m[2].conv.weight, m[2].conv.bias))(self.convs[("waveconv", i, j)])
                if chain_multi:
                    deferred.append((x, hd(1), hd(-1), hd(0) if i == 4 else None))
                    if i == 2:
                        pending = ops.head_fused_gemm_multi_nograd(deferred)
                else:
                    pending.append(ops.head_fused_gemm_nograd(x, hd(1), hd(-1), hd(0) if i == 4 else None))
                if i == 2 and pyramid:
                    continue        # ... completed inside level 1's launch (below)
                if i == 2:
                    done = ops.head_shiftsum_chain_nograd(pending, [2.0 ** (k - 1) for k in (4, 3, 2)], [1.0 / 2 ** (k - 1) for k in (4, 3, 2)],
                                                          scale_ll=2.0 ** 4)
                    for k, (yh, out, disp, yl_ll) in zip((4, 3, 2), done):
                        self.outputs[("wavelets", k - 1, "LL")] = yl_ll if k == 4 else yl
                        self.outputs[("wavelets", k - 1, "LH")] = yh[:, :, 0]
                        self.outputs[("wavelets", k - 1, "HL")] = yh[:, :, 1]
                        self.outputs[("wavelets", k - 1, "HH")] = yh[:, :, 2]
                        self.outputs[("disp", k - 1)] = disp
                        yl = out
                continue
            if pyramid and i == 1:
                hp, hn = self.convs[("waveconv", 1, 1)], self.convs[("waveconv", 1, -1)]
                hw = lambda m: (m[0].conv.weight, m[0].conv.bias, m[2].conv.weight, m[2].conv.bias)
                done, (yh, out, disp) = ops.head_level_pyramid_nograd(
                    x, hw(hp), hw(hn), 1.0, 1.0, pending, [2.0 ** (k - 1) for k in (4, 3, 2)], [1.0 / 2 ** (k - 1) for k in (4, 3, 2)],
                    scale_ll=2.0 ** 4)
                for k, (yhk, outk, dispk, yl_ll) in zip((4, 3, 2), done):
                    self.outputs[("wavelets", k - 1, "LL")] = yl_ll if k == 4 else yl
                    self.outputs[("wavelets", k - 1, "LH")] = yhk[:, :, 0]
                    self.outputs[("wavelets", k - 1, "HL")] = yhk[:, :, 1]
                    self.outputs[("wavelets", k - 1, "HH")] = yhk[:, :, 2]
                    self.outputs[("disp", k - 1)] = dispk
                    yl = outk
                self.outputs[("wavelets", 0, "LL")] = yl
                self.outputs[("wavelets", 0, "LH")] = yh[:, :, 0]
                self.outputs[("wavelets", 0, "HL")] = yh[:, :, 1]
                self.outputs[("wavelets", 0, "HH")] = yh[:, :, 2]
                self.outputs[("disp", 0)] = disp
                yl = out
                continue
            if overlap:
                keep.append(x)
                x.record_stream(side)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    yl = self._level_heads(i, x, yl)
            else:
                yl = self._level_heads(i, x, yl)
        if overlap:
            main.wait_stream(side)
        return self.outputs

    def _fused_train_ok(self, i, x):
        """Training mode: may level i's heads + synthesis run as ops.fused_level_train?  (Same conditions as the stacked form
        of get_coefficients -- whole 16-channel tiles, WMD_STACKED_HEADS -- plus what the fused kernels can write.)"""
        if not torch.is_grad_enabled() or not self.stack_heads or not self.fuse_heads:
            return False
        if any(self.convs[("waveconv", i, j)][0].conv.weight.shape[0] % 16 for j in ([0] if i == 4 else []) + [1]):
            return False
        return int(self.num_ch_dec[i]) in ops.FUSED_HEAD_WIDTHS and ops.fused_train_supported(int(self.num_ch_dec[i]), x.shape[2], x.shape[3], i == 4)

    def _level_heads(self, i, x, yl):
        """Coefficients, IDWT and disparity of level i from the trunk activation x (and the previous low-pass yl)."""
        fused = (not torch.is_grad_enabled()) and int(self.num_ch_dec[i]) in ops.FUSED_HEAD_WIDTHS and self.fuse_heads
        if fused:
            # one launch (C = 32) or two (1x1 -> LeakyReLU -> tap-partials with mid on chip, then 9-tap gather + sigmoid +
            # combine + Haar synthesis)
            hp, hn = self.convs[("waveconv", i, 1)], self.convs[("waveconv", i, -1)]
            head_ll = None
            if i == 4:   # the LL head (C -> C/4 -> 1) exists only at the coarsest level: a third chain of the same launches
                h0 = self.convs[("waveconv", i, 0)]
                head_ll = (h0[0].conv.weight, h0[0].conv.bias, h0[2].conv.weight, h0[2].conv.bias)
            yl_in = yl
            res = ops.head_fused_level_nograd(
                x, (hp[0].conv.weight, hp[0].conv.bias, hp[2].conv.weight, hp[2].conv.bias),
                (hn[0].conv.weight, hn[0].conv.bias, hn[2].conv.weight, hn[2].conv.bias),
                scale=2.0 ** (i - 1), yl=yl, disp_scale=1.0 / 2 ** (i - 1), clamp01=True, head_ll=head_ll, scale_ll=2.0 ** i)
            yh, yl, disp = res[:3]
            if head_ll is not None:
                yl_in = res[3]
            self.outputs[("wavelets", i - 1, "LL")] = yl_in
        elif self._fused_train_ok(i, x):
            # training forward on the same fused kernels (round 5, ops._FusedLevelFn): heads + synthesis of the level as ONE
            # autograd node that keeps the 1x1 outputs and the sigmoid outputs for the hand-written backward
            gate = ("elu", 0.0) if getattr(self, "_gated", False) else None
            hd = lambda j: (lambda m: (m[0].conv.weight, m[0].conv.bias, m[2].conv.weight, m[2].conv.bias))(self.convs[("waveconv", i, j)])
            yl_in = yl
            yh, yl_ll, yl, disp, mid = ops.fused_level_train(x, hd(1), hd(-1), 2.0 ** (i - 1), yl=None if i == 4 else yl,
                                                             disp_scale=1.0 / 2 ** (i - 1), clamp01=True,
                                                             head_ll=hd(0) if i == 4 else None, scale_ll=2.0 ** i, x_gate=gate)
            if i == 4:
                yl_in = yl_ll
            if self.branch_trace is not None:   # which LeakyReLU piece each element took (gradient-parity diagnostics)
                o = 0
                for j in ([0] if i == 4 else []) + [1, -1]:
                    c = self.convs[("waveconv", i, j)][0].conv.weight.shape[0]
                    self.branch_trace[("waveconv", i, j)] = (mid[:, o:o + c] > 0).cpu()
                    o += c
            self.outputs[("wavelets", i - 1, "LL")] = yl_in
            yh = yh.unsqueeze(1)
            fused = True
        else:
            gate = ("elu", 0.0) if (torch.is_grad_enabled() and getattr(self, "_gated", False)) else None   # x is this decoder's own ELU output, its producer expects dz (see _forward_impl)
            if i == 4:
                yl, yh = self.get_coefficients(x, scale=i, return_ll=True, _x_gate=gate)
            else:
                _, yh = self.get_coefficients(x, scale=i, return_ll=False, _x_gate=gate)
            self.outputs[("wavelets", i - 1, "LL")] = yl
        self.outputs[("wavelets", i - 1, "LH")] = yh[:, :, 0]
        self.outputs[("wavelets", i - 1, "HL")] = yh[:, :, 1]
        self.outputs[("wavelets", i - 1, "HH")] = yh[:, :, 2]
        if not fused:
            yl, disp = ops.idwt_haar(yl, yh, disp_scale=1.0 / 2 ** (i - 1), clamp01=True)
        self.outputs[("disp", i - 1)] = disp
        return yl


class DepthDecoder(nn.Module):
    """Non-wavelet baseline (reference :18-69): zero-padded ConvBlocks, reflect-padded dispconv + sigmoid."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            num_ch_in = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
            num_ch_in = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                num_ch_in += self.num_ch_enc[i - 1]
            self.convs[("upconv", i, 1)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features):
        input_features, edge = split_edge(input_features)
        if edge is not None:
            input_features[-1] = edge.activate()      # the baseline decoder takes the activated tensor (no fused edge)
        self.outputs = {}
        ops.prepack_module(self)
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            skip = input_features[i - 1] if (self.use_skips and i > 0) else None
            x = self.convs[("upconv", i, 1)](x, skip=skip, up=2)
            if i in self.scales:
                c = self.convs[("dispconv", i)]
                if self.num_output_channels <= 4:
                    self.outputs[("disp", i)] = ops.head3x3(x, c.conv.weight, c.conv.bias, pad="reflect", mode=1, scale=1.0)
                else:
                    self.outputs[("disp", i)] = c(x, act="sigmoid")
        return self.outputs
