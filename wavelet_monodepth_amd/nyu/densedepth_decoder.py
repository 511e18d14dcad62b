"""NYUv2 DenseDepth-style wavelet decoders on MI355X, API-compatible with the reference
(/root/reference/NYUv2/networks/decoders/densedepth_decoder.py):

  DecoderWave        :92-148    dense
  SparseDecoderWave  :224-409   two sparse levels (see sparse_decoder.py)

Same constructors, `state_dict` names (conv2.conv.*, up{1,2,3}.convA.conv.*, wave1_ll.conv.*, wave{1,2,3}.conv.*,
iwt.*, iwt_LL.*) and output keys.  The depthwise options (dw_waveconv / dw_upconv) and the baseline
Decoder/Decoder224/DecoderWave224 classes are SURVEY §8(f) "next" items.
"""
import torch
import torch.nn as nn

from .. import ops
from ..layers import NyuConv3x3, UpSampleBlock
from ..wavelets import IDWT
from ..graphs import GraphCache


class DecoderWave(nn.Module):
    def __init__(self, enc_features=[96, 96, 192, 384, 2208], decoder_width=0.5, dw_waveconv=False, dw_upconv=False):
        super().__init__()
        features = int(enc_features[-1] * decoder_width)
        wave_pad = "zero"
        padding = "reflection"
        self.iwt = IDWT(wave="haar", mode=wave_pad)
        self.iwt_LL = IDWT(wave="haar", mode="zero")
        self.conv2 = NyuConv3x3(enc_features[-1], features, padding="replicate")
        self.up1 = UpSampleBlock(skip_input=features // 1 + enc_features[-2], output_features=features // 2,
                                 padding=padding, is_depthwise=dw_upconv)
        self.wave1_ll = NyuConv3x3(features // 2, 1, padding="replicate")
        self.wave1 = NyuConv3x3(features // 2, 3, padding=wave_pad, is_depthwise=dw_waveconv)
        self.up2 = UpSampleBlock(skip_input=features // 2 + enc_features[-3], output_features=features // 4,
                                 padding=padding, is_depthwise=dw_upconv)
        self.wave2 = NyuConv3x3(features // 4, 3, padding=wave_pad, is_depthwise=dw_waveconv)
        self.up3 = UpSampleBlock(skip_input=features // 4 + enc_features[-4], output_features=features // 8,
                                 padding=padding, is_depthwise=dw_upconv)
        self.wave3 = NyuConv3x3(features // 8, 3, padding=wave_pad, is_depthwise=dw_waveconv)
        self._graph_mode = False
        self._graphs = GraphCache()

    @staticmethod
    def _wave(conv, x, scale):
        """scale * Conv3x3(C, 1|3)(x) through the small-Cout head kernel."""
        return ops.head3x3(x, conv.conv.weight, conv.conv.bias, pad=conv.pad_mode, mode=0, scale=scale)

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
        x_d0 = self.conv2(x_blocks[-1])
        x_d1 = self.up1(x_d0, x_blocks[-2])
        ll = self._wave(self.wave1_ll, x_d1, 2.0 ** 3)
        outputs[("disp", 3)] = ll / (2 ** 3)
        h = self._wave(self.wave1, x_d1, 2.0 ** 2).unsqueeze(1)
        outputs[("wavelets", 2, "LL")] = ll
        outputs[("wavelets", 2, "LH")] = h[:, :, 0]
        outputs[("wavelets", 2, "HL")] = h[:, :, 1]
        outputs[("wavelets", 2, "HH")] = h[:, :, 2]
        ll, disp = ops.idwt_haar(ll, h, disp_scale=1.0 / 2 ** 2, clamp01=False)
        outputs[("disp", 2)] = disp

        x_d2 = self.up2(x_d1, x_blocks[-3])
        h = self._wave(self.wave2, x_d2, 2.0 ** 1).unsqueeze(1)
        outputs[("wavelets", 1, "LH")] = h[:, :, 0]
        outputs[("wavelets", 1, "HL")] = h[:, :, 1]
        outputs[("wavelets", 1, "HH")] = h[:, :, 2]
        ll, disp = ops.idwt_haar(ll, h, disp_scale=1.0 / 2 ** 1, clamp01=False)
        outputs[("disp", 1)] = disp

        x_d3 = self.up3(x_d2, x_blocks[-4])
        h = self._wave(self.wave3, x_d3, 1.0).unsqueeze(1)
        outputs[("wavelets", 0, "LH")] = h[:, :, 0]
        outputs[("wavelets", 0, "HL")] = h[:, :, 1]
        outputs[("wavelets", 0, "HH")] = h[:, :, 2]
        ll, _ = ops.idwt_haar(ll, h)
        outputs[("disp", 0)] = ll
        return outputs
