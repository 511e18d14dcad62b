"""The NYUv2 project's layer names (/root/reference/NYUv2/networks/layers.py:11-67): `Conv3x3(in_channels, out_channels,
padding="zero", stride=1, is_depthwise=False)` and `UpSampleBlock(skip_input, output_features, padding="zero",
is_depthwise=False)`, implemented in wavelet_monodepth_amd/layers.py (where the KITTI flavour owns the name Conv3x3)."""
from ..layers import NyuConv3x3 as Conv3x3, UpSampleBlock  # noqa: F401
