"""`Model(opts)` of the reference's NYUv2 project (/root/reference/NYUv2/model.py:12-71): encoder + decoder picked from
the option namespace, `forward(x, threshold=-1)` hands the threshold to the sparse decoder only.

The decoders are this package's HIP-backed classes.  The encoder stays ordinary PyTorch (BASELINE.json north_star): the
reference takes DenseNet / ResNet / MobileNetV2 from torchvision, which this image does not have, so
  * `encoder_type == "resnet"` builds the plain-torch ResNet of `wavelet_monodepth_amd.encoders` (torchvision-compatible
    state_dict names, five feature maps, `num_ch_enc` as resnet_encoder.py:66-91), and
  * any other encoder is passed in ready-made (`Model(opts, encoder=DenseEncoder(...))`): a module whose forward returns
    the five feature maps and that exposes `num_ch_enc` — exactly what the reference's encoder classes provide.
Nothing is printed (the reference prints "Building model using ... encoder").
"""
import numpy as np
import torch.nn as nn

from ..encoders import _ResNet, _SPECS
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


class Model(nn.Module):
    def __init__(self, opts, encoder=None):
        super().__init__()
        decoder_width = 0.5
        if encoder is not None:
            self.encoder = encoder
        elif opts.encoder_type == "resnet":
            self.encoder = NyuResnetEncoder(num_layers=opts.num_layers, pretrained=opts.pretrained_encoder,
                                            normalize_input=opts.normalize_input)
        elif opts.encoder_type in ("densenet", "mobilenet", "mobilenet_light"):
            raise NotImplementedError("the %s encoder comes from torchvision, which is not part of this package: build it "
                                      "and pass it as Model(opts, encoder=...)" % opts.encoder_type)
        else:
            raise NotImplementedError

        self.use_sparse = False
        if opts.use_wavelets:
            # model.py:37-46: a namespace without `use_sparse` means dense; sparse + 224 is refused
            try:
                if opts.use_sparse:
                    self.use_sparse = True
                    if opts.use_224:
                        raise NotImplementedError
            except AttributeError:
                opts.use_sparse = False
                self.use_sparse = False
            if opts.use_sparse:
                self.decoder = SparseDecoderWave(enc_features=self.encoder.num_ch_enc, decoder_width=decoder_width)
            else:
                decoder_wave = DecoderWave224 if opts.use_224 else DecoderWave
                self.decoder = decoder_wave(enc_features=self.encoder.num_ch_enc, decoder_width=decoder_width,
                                            dw_waveconv=opts.dw_waveconv, dw_upconv=opts.dw_upconv)
        else:
            decoder = Decoder224 if opts.use_224 else Decoder
            self.decoder = decoder(enc_features=self.encoder.num_ch_enc,
                                   is_depthwise=(opts.dw_waveconv or opts.dw_upconv))

    def forward(self, x, threshold=-1):
        x = self.encoder(x)
        if self.use_sparse:
            return self.decoder(x, threshold)
        return self.decoder(x)
