"""Import shim used ONLY by tests/golden/make_golden.py inside the build container.

The reference imports `from pytorch_wavelets import IDWT`, a third-party package that is neither
vendored in /root/reference nor installed in this image.  To let the reference's own decoder
modules run unmodified, this shim provides an `IDWT` whose arithmetic is *the reference's own*
closed-form Haar synthesis, `SparseDepthWaveProgressiveDecoder.my_iwt_once`
(/root/reference/KITTI/networks/decoders/depth_decoder.py:225-239), loaded from the reference file
at call time.  No wavelet arithmetic is written here.  It never travels to the GPU box and is not
imported by the package, the tests or the benchmark.
"""
import importlib.util
import os
import sys

import torch.nn as nn

_REF_KITTI = "/root/reference/KITTI"
_iwt = None


def _reference_iwt():
    global _iwt
    if _iwt is None:
        mod = sys.modules.get("networks.decoders.depth_decoder")
        if mod is None or not hasattr(mod, "SparseDepthWaveProgressiveDecoder"):
            # NYUv2 session: `networks` is the NYUv2 package, so load the KITTI file under another name
            sys.path.insert(0, _REF_KITTI)
            try:
                spec = importlib.util.spec_from_file_location(
                    "_kitti_depth_decoder_for_iwt", os.path.join(_REF_KITTI, "networks/decoders/depth_decoder.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
            finally:
                sys.path.remove(_REF_KITTI)
        _iwt = mod.SparseDepthWaveProgressiveDecoder.my_iwt_once
    return _iwt


class IDWT(nn.Module):
    def __init__(self, wave="haar", mode="zero"):
        super().__init__()
        assert wave == "haar", "the reference only ever asks for Haar"
        self.mode = mode

    def forward(self, coeffs):
        yl, yh = coeffs
        assert len(yh) == 1, "the reference only ever inverts one level at a time"
        return _reference_iwt()((yl, [yh[0]]))


DWTInverse = IDWT
