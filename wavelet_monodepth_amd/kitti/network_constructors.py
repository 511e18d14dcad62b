"""Factories with the reference's names, signatures and selection logic
(/root/reference/KITTI/networks/network_constructors.py): `make_depth_decoder(encoder, opts)` (:30-40) and
`make_depth_encoder(opts)` (:12-27; the encoders are the plain torch.nn ones of `wavelet_monodepth_amd.encoders`).  The pose factory of that file is out of scope.  Nothing is printed.

`make_depth_decoder` also accepts the explicit form `make_depth_decoder(num_ch_enc, scales, use_wavelets=..., use_sparse=...)`
for callers that have no option namespace.
"""
from .depth_decoder import DepthDecoder, DepthWaveProgressiveDecoder


def make_depth_encoder(opts):
    from ..encoders import MobileNetV2Encoder, ResnetEncoder
    if opts.encoder_type == "resnet":
        # weights_init == "pretrained" downloads ImageNet weights in the reference; there is no network here, so the
        # encoder refuses `pretrained=True` with a clear message and checkpoints are loaded explicitly instead
        return ResnetEncoder(opts.num_layers, pretrained=(opts.weights_init == "pretrained"))
    if opts.encoder_type in ("mobilenet", "mobilenet_light"):
        return MobileNetV2Encoder(pretrained=(opts.weights_init == "pretrained"), use_last_layer=(opts.encoder_type == "mobilenet"))
    raise NotImplementedError


def make_depth_decoder(encoder, opts=None, use_wavelets=False, use_sparse=False):
    if hasattr(encoder, "num_ch_enc") and opts is not None and hasattr(opts, "use_wavelets"):
        # the reference's call: make_depth_decoder(encoder, opts)  (trainer.py:72, test_simple.py:97)
        num_ch_enc, scales = encoder.num_ch_enc, opts.scales
        use_wavelets, use_sparse = opts.use_wavelets, opts.use_sparse
    else:
        num_ch_enc, scales = encoder, (range(4) if opts is None else opts)
    if use_wavelets:
        if use_sparse:
            from .sparse_decoder import SparseDepthWaveProgressiveDecoder
            # the reference builds the sparse decoder with its default scales (network_constructors.py:34)
            return SparseDepthWaveProgressiveDecoder(num_ch_enc) if hasattr(encoder, "num_ch_enc") \
                else SparseDepthWaveProgressiveDecoder(num_ch_enc, scales)
        return DepthWaveProgressiveDecoder(num_ch_enc, scales)
    return DepthDecoder(num_ch_enc, scales)
