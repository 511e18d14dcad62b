"""`Model(opts)` of the reference's NYUv2 project (/root/reference/NYUv2/model.py:12-71): encoder + decoder picked from
the option namespace, `forward(x, threshold=-1)` hands the threshold to the sparse decoder only.

The decoders are this package's HIP-backed classes.  The encoder stays ordinary PyTorch (BASELINE.json north_star): the
reference takes DenseNet / ResNet / MobileNetV2 from torchvision, which this image does not have, so
`wavelet_monodepth_amd.encoders` defines them in plain torch.nn with torchvision-compatible state_dict names
(`densenet` = DenseNet161 -> num_ch_enc [96, 96, 192, 384, 2208]; `mobilenet` / `mobilenet_light` = MobileNetV2 with / without
the 1280-channel last layer).  A ready-made encoder can also be handed in (`Model(opts, encoder=...)`): any module whose
forward returns the five feature maps and that exposes `num_ch_enc`.
Nothing is printed (the reference prints "Building model using ... encoder"), and the option namespace is only read.
"""
import numpy as np
import torch.nn as nn

from ..encoders import DenseEncoder, MobileNetV2Encoder, _ResNet, _SPECS
from .densedepth_decoder import Decoder, Decoder224, DecoderWave, DecoderWave224, SparseDecoderWave


class NyuResnetEncoder(nn.Module):
    """NYUv2/networks/encoders/resnet_encoder.py:62-106: like the KITTI encoder but WITHOUT input normalisation — the
    reference's `normalize_input` loop (`t.sub(m).div(s)`, :95-97) discards its result, so the flag is accepted and,
    as in the reference, changes nothing."""

    def __init__(self, num_layers=18, pretrained=False, num_input_images=1, normalize_input=False):
        super().__init__()
        if pretrained:
            raise RuntimeError("no network access in this environment: load the encoder weights explicitly")
        if num_layers not in _SPECS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        block, layers = _SPECS[num_layers]
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self.normalize_input = normalize_input
        self.encoder = _ResNet(block, layers, num_input_images)

    def forward(self, input_image):
        e = self.encoder
        f0 = e.relu(e.bn1(e.conv1(input_image)))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]


def _build_encoder(opts):
    kind = opts.encoder_type
    if kind == "resnet":
        return NyuResnetEncoder(num_layers=opts.num_layers, pretrained=opts.pretrained_encoder,
                                normalize_input=opts.normalize_input)
    if kind == "densenet":
        return DenseEncoder(normalize_input=opts.normalize_input, pretrained=opts.pretrained_encoder)
    if kind in ("mobilenet", "mobilenet_light"):
        return MobileNetV2Encoder(pretrained=opts.pretrained_encoder, use_last_layer=(kind == "mobilenet"),
                                  normalize_input=opts.normalize_input)
    raise NotImplementedError("encoder_type %r" % (kind,))


# (use_wavelets, use_224, use_sparse) -> decoder class; a missing row is a combination the reference refuses (model.py:41)
_DECODERS = {
    (False, False): Decoder,
    (False, True): Decoder224,
    (True, False, False): DecoderWave,
    (True, True, False): DecoderWave224,
    (True, False, True): SparseDecoderWave,
}


class Model(nn.Module):
    DECODER_WIDTH = 0.5

    def __init__(self, opts, encoder=None):
        super().__init__()
        self.encoder = _build_encoder(opts) if encoder is None else encoder
        wave, w224 = bool(opts.use_wavelets), bool(opts.use_224)
        depthwise = dict(dw_waveconv=opts.dw_waveconv, dw_upconv=opts.dw_upconv)
        chans = self.encoder.num_ch_enc
        if not wave:
            # the sparse switch only exists for the wavelet decoders (model.py:36-47 never looks at it otherwise)
            self.use_sparse = False
            self.decoder = _DECODERS[(False, w224)](enc_features=chans, is_depthwise=any(depthwise.values()))
            return
        self.use_sparse = bool(getattr(opts, "use_sparse", False))    # namespaces written before the option existed: dense
        cls = _DECODERS.get((True, w224, self.use_sparse))
        if cls is None:
            raise NotImplementedError("there is no sparse 224x224 wavelet decoder")
        if self.use_sparse:
            self.decoder = cls(enc_features=chans, decoder_width=self.DECODER_WIDTH)
        else:
            self.decoder = cls(enc_features=chans, decoder_width=self.DECODER_WIDTH, **depthwise)

    def forward(self, x, threshold=-1):
        feats = self.encoder(x)
        return self.decoder(feats, threshold) if self.use_sparse else self.decoder(feats)
