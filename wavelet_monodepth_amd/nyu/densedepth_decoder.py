"""NYUv2 DenseDepth-style wavelet decoders on MI355X, API-compatible with the reference
(/root/reference/NYUv2/networks/decoders/densedepth_decoder.py):

  Decoder            :15-46     DenseDepth baseline (no wavelets), zero padding
  Decoder224         :49-89     224x224 variant: one more nearest x2 + Conv3x3/LeakyReLU before the output conv
  DecoderWave        :92-148    dense
  DecoderWave224     :151-221   four wavelet levels (incl. the reference's floor division of ("disp", 1), :212)
  SparseDecoderWave  :224-409   two sparse levels (see sparse_decoder.py)

Same constructors, `state_dict` names (conv2.conv.*, up{1..4}.convA.conv.*, conv5.0.conv.*, conv3.*, wave1_ll.conv.*,
wave{1..4}.conv.*, iwt.*, iwt_LL.*) and output keys, incl. the depthwise options (is_depthwise / dw_waveconv / dw_upconv:
depthwise 3x3 -> ReLU -> 1x1, keys *.conv.0.0.weight / *.conv.1.weight).
"""
import torch
import torch.nn as nn

from .. import ops, sparse_ops as S
from ..layers import NyuConv3x3, UpSampleBlock, gated_backward_allowed
from ..wavelets import IDWT
from ..graphs import GraphCache


class _OutConv3x3(nn.Conv2d):
    """The baselines' output layer is a bare nn.Conv2d(C, 1, 3, padding=1) (densedepth_decoder.py:33,71): same
    parameter names (conv3.weight / conv3.bias), executed by the small-Cout head kernel with zero padding."""

    def __init__(self, in_channels, out_channels):
        super().__init__(int(in_channels), int(out_channels), kernel_size=3, stride=1, padding=1, padding_mode="zeros")

    def forward(self, x):
        return ops.head3x3(x, self.weight, self.bias, pad="zero", mode=0, scale=1.0)


def _add_levels(dec, features, enc_features, n_up, padding, dw_up=False, wave_pad=None, dw_wave=False):
    """Registers up1..up<n_up> (densedepth_decoder.py:23-31,103-114,166-182: level k joins the decoder width halved k - 1
    times with the encoder block k + 1 from the end and halves the width again) and, for the wavelet decoders, wave1_ll
    after up1 and wave<k> after up<k> -- in the reference's registration order, so `state_dict()` lists the same keys."""
    for k in range(1, n_up + 1):
        setattr(dec, "up%d" % k, UpSampleBlock(skip_input=features // 2 ** (k - 1) + enc_features[-1 - k],
                                               output_features=features // 2 ** k, padding=padding, is_depthwise=dw_up))
        if wave_pad is not None:
            if k == 1:
                dec.wave1_ll = NyuConv3x3(features // 2, 1, padding="replicate")
            setattr(dec, "wave%d" % k, NyuConv3x3(features // 2 ** k, 3, padding=wave_pad, is_depthwise=dw_wave))


def _log_highs(outputs, scale, h):
    for j, band in enumerate(("LH", "HL", "HH")):
        outputs[("wavelets", scale, band)] = h[:, :, j]


class Decoder(nn.Module):
    def __init__(self, enc_features=[96, 96, 192, 384, 2208], decoder_width=0.5, is_depthwise=False):
        super().__init__()
        features = int(enc_features[-1] * decoder_width)
        padding = "zero"
        self.conv2 = NyuConv3x3(enc_features[-1], features, padding="zero")
        _add_levels(self, features, enc_features, 4, padding, dw_up=is_depthwise)
        # (:32-35) a bare nn.Conv2d unless the depthwise option is on
        self.conv3 = NyuConv3x3(features // 16, 1, is_depthwise=True) if is_depthwise else _OutConv3x3(features // 16, 1)

    def _trunk(self, features):
        ops.prepack_module(self)
        x_block0, x_block1, x_block2, x_block3, x_block4 = tuple(features)
        x_d0 = self.conv2(x_block4)
        x_d1 = self.up1(x_d0, x_block3)
        x_d2 = self.up2(x_d1, x_block2)
        x_d3 = self.up3(x_d2, x_block1)
        return self.up4(x_d3, x_block0)

    def forward(self, features):
        return {("disp", 0): self.conv3(self._trunk(features))}


class Decoder224(Decoder):
    def __init__(self, enc_features=[96, 96, 192, 384, 2208], decoder_width=0.5, is_depthwise=False):
        super().__init__(enc_features=enc_features, decoder_width=decoder_width, is_depthwise=is_depthwise)
        features = int(enc_features[-1] * decoder_width)
        # nn.Sequential(Conv3x3, LeakyReLU(0.2)) in the reference (:66-67): key conv5.0.conv.*; the nearest x2 in front of
        # it (:87) and the activation are fused into the convolution
        self.conv5 = nn.Sequential(NyuConv3x3(features // 16, features // 32, is_depthwise=is_depthwise), nn.LeakyReLU(0.2))
        self.conv3 = NyuConv3x3(features // 32, 1, is_depthwise=True) if is_depthwise else _OutConv3x3(features // 32, 1)

    def forward(self, features):
        x_d5 = self.conv5[0](self._trunk(features), up=2, act="leaky", slope=0.2)
        return {("disp", 0): self.conv3(x_d5)}


class DecoderWave(nn.Module):
    def __init__(self, enc_features=[96, 96, 192, 384, 2208], decoder_width=0.5, dw_waveconv=False, dw_upconv=False):
        super().__init__()
        features = int(enc_features[-1] * decoder_width)
        wave_pad = "zero"
        padding = "reflection"
        self.iwt = IDWT(wave="haar", mode=wave_pad)
        self.iwt_LL = IDWT(wave="haar", mode="zero")
        self.conv2 = NyuConv3x3(enc_features[-1], features, padding="replicate")
        _add_levels(self, features, enc_features, 3, padding, dw_up=dw_upconv, wave_pad=wave_pad, dw_wave=dw_waveconv)
        self._graph_mode = False
        self._graphs = GraphCache()

    @staticmethod
    def _wave(conv, x, scale):
        """scale * Conv3x3(C, 1|3)(x): the small-Cout head kernel, or depthwise + 1x1 for the dw_waveconv option."""
        return conv.head(x, scale)

    def enable_graph(self, on=True):
        self._graph_mode = bool(on)
        self._graphs.clear()
        return self

    def forward(self, x_blocks):
        if self._graph_mode and not torch.is_grad_enabled():
            return self._graphs.run(self._forward_impl, x_blocks, self.parameters())
        return self._forward_impl(x_blocks)

    def _forward_impl(self, x_blocks):
        outputs = {}
        ops.prepack_module(self)
        # training (plain 3x3 layers): every consumer of an UpSampleBlock output -- the next block and the wavelet heads --
        # returns its data gradient multiplied by LeakyReLU(0.2)'(output), so no block runs a separate activation-backward
        # pass (ops.conv2d_fused: x1_gate / grad_is_dz)
        plain = not (self.up1.convA.is_depthwise or self.wave1.is_depthwise)
        g = ("leaky", 0.2) if (torch.is_grad_enabled() and plain and gated_backward_allowed(self)) else None
        dz = g is not None
        x = self.up1(self.conv2(x_blocks[-1]), x_blocks[-2], grad_is_dz=dz)
        ll = self.wave1_ll.head(x, 2.0 ** 3, x_gate=g)
        outputs[("disp", 3)] = ll / (2 ** 3)
        for k in (1, 2, 3):           # level k: highs of scale s = 3 - k, scaled 2^s (:122-147); no sigmoid, no clamp
            s = 3 - k
            if k > 1:
                x = getattr(self, "up%d" % k)(x, x_blocks[-1 - k], x1_gate=g, grad_is_dz=dz)
            h = getattr(self, "wave%d" % k).head(x, 2.0 ** s, x_gate=g).unsqueeze(1)
            if k == 1:
                outputs[("wavelets", 2, "LL")] = ll
            _log_highs(outputs, s, h)
            if s:
                ll, outputs[("disp", s)] = ops.idwt_haar(ll, h, disp_scale=1.0 / 2 ** s, clamp01=False)
            else:
                ll, _ = ops.idwt_haar(ll, h)
                outputs[("disp", 0)] = ll
        return outputs


class DecoderWave224(nn.Module):
    """densedepth_decoder.py:151-221: four levels; LL head scaled by 2^4, highs by 2^3, 2^2, 2^1, 1; ("disp", s) = LL_s / 2^s
    except ("disp", 1), which the reference computes with floor division (`ll // 2`, :212) -- reproduced as is."""

    def __init__(self, enc_features=[96, 96, 192, 384, 2208], decoder_width=0.5, dw_waveconv=False, dw_upconv=False):
        super().__init__()
        features = int(enc_features[-1] * decoder_width)
        wave_pad = "zero"
        padding = "reflection"
        self.iwt = IDWT(wave="haar", mode=wave_pad)
        self.iwt_LL = IDWT(wave="haar", mode="zero")
        self.conv2 = NyuConv3x3(enc_features[-1], features, padding="replicate")
        _add_levels(self, features, enc_features, 4, padding, dw_up=dw_upconv, wave_pad=wave_pad, dw_wave=dw_waveconv)
        self.sigmoid = nn.Sigmoid()

    _wave = staticmethod(DecoderWave._wave)

    def forward(self, x_blocks):
        outputs = {}
        ops.prepack_module(self)
        x = self.up1(self.conv2(x_blocks[-1]), x_blocks[-2])
        ll = self._wave(self.wave1_ll, x, 2.0 ** 4)
        for level, (wave, up, skip) in enumerate(((self.wave1, self.up2, x_blocks[-3]), (self.wave2, self.up3, x_blocks[-4]),
                                                  (self.wave3, self.up4, x_blocks[-5]), (self.wave4, None, None))):
            s = 3 - level
            h = self._wave(wave, x, 2.0 ** s).unsqueeze(1)
            if level == 0:
                outputs[("wavelets", 3, "LL")] = ll
            _log_highs(outputs, s, h)
            if s == 1:
                ll, _ = ops.idwt_haar(ll, h)
                outputs[("disp", 1)] = ll // (2 ** 1)     # sic (floor_divide: not differentiable, as in the reference)
            elif s == 0:
                ll, _ = ops.idwt_haar(ll, h)
                outputs[("disp", 0)] = ll
            else:
                ll, outputs[("disp", s)] = ops.idwt_haar(ll, h, disp_scale=1.0 / 2 ** s, clamp01=False)
            if up is not None:
                x = up(x, skip)
        return outputs


def _sparse_conv_ops(cin, cout, nnz_out):
    # ops returned by sparse_conv3x3 (NYUv2/networks/layers.py: gathered elements + (1 + 9 cin) * nnz * cout)
    return cin * 9 * nnz_out + (1 + 9 * cin) * nnz_out * cout


def nyu_sparse_total_ops(c_in, hw_in, c_d0, c_skip1, c_d1, hw1, levels, counts):
    """The reference's op model of SparseDecoderWave.forward (densedepth_decoder.py:296-409) as a host function:
    c_in/hw_in = channels and (h, w) of the deepest encoder block, c_d0 = conv2 width, c_skip1 = channels of the first
    skip, c_d1 / hw1 = width and (h, w) of up1's output, levels = [(Cin_convA, Cout_convA, Cin_wave), ...] of the two sparse levels,
    counts = [(n_wave, n_wavelet), ...] their pixel counts.  Returns the python int `total_ops`."""
    (h0, w0), (h1, w1) = hw_in, hw1
    n = (1 + 9 * c_in) * h0 * w0 * c_d0                       # conv2 (counted with 9 taps by the reference, :301)
    n += (1 + 9 * (c_d0 + c_skip1)) * h1 * w1 * c_d1          # up1
    n += (1 + 9 * c_d1) * h1 * w1 * 4                         # wave1_ll + wave1
    n += (2 * h1) * (2 * w1)                                  # IDWT
    mh, mw = h1, w1
    for level, ((cin_a, ca, cin_w), (n_wave, n_wl)) in enumerate(zip(levels, counts)):
        n += 3 * mh * mw                                      # threshold
        n += 25 * mh * mw + 100 * mh * mw                     # dilations
        # mask2idxmap calls: wavelet, conva, wave (fine) + up (coarse) (+ a repeated `wave` at the 2nd level, :374-375)
        n += 3 * 4 * mh * mw + mh * mw + (4 * mh * mw if level == 1 else 0)
        n += (4 * mh) * (4 * mw)                              # IDWT of the level
        n += _sparse_conv_ops(cin_a, ca, n_wave) + _sparse_conv_ops(cin_w, 3, n_wl)
        mh, mw = 2 * mh, 2 * mw
    return n


class SparseDecoderWave(DecoderWave):
    """SparseDecoderWave (reference densedepth_decoder.py:224-409): dense down to scale 2, then two
    threshold-gated levels that only use up{2,3}.convA and wave{2,3}.  `forward(x_blocks, thresh_ratio=0.1)`,
    batch 1, inference only; returns the reference's keys incl. ("wavelet_mask", s) and "total_ops".
    (The reference prints "Using Sparse DenseDepth Decoder" on construction; this class does not.)"""

    def __init__(self, enc_features=[96, 96, 192, 384, 2208], decoder_width=0.5):
        super().__init__(enc_features=enc_features, decoder_width=decoder_width)

    @torch.no_grad()
    def forward(self, x_blocks, thresh_ratio=0.1, _force_masks=None):
        out = {}
        xb = x_blocks
        assert xb[-1].shape[0] == 1, "works with single input only"
        dev = xb[-1].device
        w2 = self.conv2.conv.weight
        x_d0 = self.conv2(xb[-1])
        x_d1 = self.up1(x_d0, xb[-2])
        ll = self._wave(self.wave1_ll, x_d1, 2.0 ** 3)
        out[("disp", 3)] = ll / (2 ** 3)
        h = self._wave(self.wave1, x_d1, 2.0 ** 2).unsqueeze(1)
        out[("wavelet_mask", 2)] = torch.ones_like(h[:, 0])
        out[("wavelets", 2, "LL")] = ll
        _log_highs(out, 2, h)
        ll, disp = ops.idwt_haar(ll, h, disp_scale=0.25, clamp01=False)
        out[("disp", 2)] = disp

        src = x_d1[0].contiguous()       # dense [C,h,w]; later: the previous level's convA output buffer
        pending = []
        for level, (up, wave, skip, scale) in enumerate(((self.up2, self.wave2, xb[-3], 2.0), (self.up3, self.wave3, xb[-4], 1.0))):
            mh, mw = src.shape[-2:]
            specs = [(1, 2), (2, 2), (2, 1), (2, 0)]
            if _force_masks is not None and level in _force_masks:
                mask = _force_masks[level].to(dev).reshape(mh, mw).to(torch.uint8).contiguous()
                up_mask, conva_mask, wave_mask, wavelet_mask = S.dilate_multi(mask, specs)
            else:   # min/max + threshold + the four dilations in one launch
                up_mask, conva_mask, wave_mask, wavelet_mask = S.mask_level(ll, h, thresh_ratio, specs)
            H2, W2 = 2 * mh, 2 * mw
            s = 1 - level
            out[("wavelet_mask", s)] = wavelet_mask.to(torch.float32).reshape(1, 1, H2, W2)
            (co_wave, co_wl), nnz = S.compact_multi([wave_mask, wavelet_mask])
            ca = up.convA.conv
            Ca = ca.weight.shape[0]
            xa = torch.zeros((Ca, H2, W2), device=dev)
            # sparse_upsample (coarse values on up_mask are a subset test away: conva_mask lies inside up2(up_mask))
            # + convA + LeakyReLU(0.2), reflect index padding, outputs on wave_mask
            S.sparse_conv(xa, src, ops.pack_weights(ca.weight), ca.bias, Ca, 3, co_wave, nnz.data_ptr(), H2 * W2,
                          x2=skip[0].contiguous(), up1=2, in_mask=conva_mask, pad="reflect", act="leaky", slope=0.2)
            cw = wave.conv
            hd = torch.zeros((1, 3, H2, W2), device=dev)
            S.sparse_conv(hd[0], xa, ops.pack_weights(cw.weight), cw.bias, 3, 3, co_wl, nnz.data_ptr() + 4, H2 * W2,
                          in_mask=wave_mask, pad="zero", act="none", out_scale=scale)
            h = hd.unsqueeze(1)
            _log_highs(out, s, h)
            ll, disp = ops.idwt_haar(ll, h, disp_scale=0.5 if level == 0 else 1.0, clamp01=False)
            out[("disp", s)] = disp if level == 0 else ll
            pending.append((nnz, ca.weight.shape[1], Ca, cw.weight.shape[1]))
            src = xa
        # the python-int op model is resolved on first access from counts copied to pinned host memory (no sync here)
        out = S.LazyOpsDict(out)
        fetch = S.counts_to_host([nnz for nnz, *_ in pending])
        shapes = (xb[-1].shape[1], tuple(xb[-1].shape[2:]), w2.shape[0], xb[-2].shape[1], x_d1.shape[1], tuple(x_d1.shape[2:]),
                  [(cin_a, ca_, cin_w) for _nnz, cin_a, ca_, cin_w in pending])
        out.set_lazy(["total_ops"], lambda: {"total_ops": nyu_sparse_total_ops(*shapes, [tuple(c) for c in fetch()])})
        return out
