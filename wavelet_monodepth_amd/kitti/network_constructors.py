"""Decoder factory with the reference's selection logic
(/root/reference/KITTI/networks/network_constructors.py:30-40: `make_depth_decoder`).
The encoder / pose factories of that file are out of scope; a plain-torch
ResNet encoder for end-to-end runs lives in wavelet_monodepth_amd/encoders.py."""
from .depth_decoder import DepthDecoder, DepthWaveProgressiveDecoder


def make_depth_decoder(num_ch_enc, scales, use_wavelets=False, use_sparse=False):
    if use_wavelets:
        if use_sparse:
            from .sparse_decoder import SparseDepthWaveProgressiveDecoder
            return SparseDepthWaveProgressiveDecoder(num_ch_enc, scales)
        return DepthWaveProgressiveDecoder(num_ch_enc, scales)
    return DepthDecoder(num_ch_enc, scales)
