"""Drop-in for the two pytorch_wavelets classes the reference imports (`IDWT`, `DWT`), Haar only,
running on libwmd_hip.so.  Call sites: KITTI/networks/decoders/depth_decoder.py:85,164;
NYUv2/networks/decoders/densedepth_decoder.py:99-101; NYUv2/train.py:258,289."""
import math

import torch
import torch.nn as nn

from . import ops


class IDWT(nn.Module):
    """`IDWT(wave="haar", mode="zero")`: forward((yl, [yh])) -> yl_next, one level per entry of yh.
    Registers the four synthesis-filter buffers upstream DWTInverse keeps (names recalled from
    pytorch_wavelets 1.3.0) so that released checkpoints load with strict=True; their values are not
    used by the kernel."""

    def __init__(self, wave="haar", mode="zero"):
        super().__init__()
        if wave != "haar":
            raise NotImplementedError("only the Haar wavelet is used by the reference")
        self.mode = mode
        r = 1.0 / math.sqrt(2.0)
        self.register_buffer("g0_col", torch.tensor([r, r]).reshape(1, 1, 2, 1))
        self.register_buffer("g1_col", torch.tensor([r, -r]).reshape(1, 1, 2, 1))
        self.register_buffer("g0_row", torch.tensor([r, r]).reshape(1, 1, 1, 2))
        self.register_buffer("g1_row", torch.tensor([r, -r]).reshape(1, 1, 1, 2))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # tolerate any upstream buffer naming: the filters are constants
        for k in [k for k in state_dict if k.startswith(prefix)]:
            name = k[len(prefix):]
            if name not in ("g0_col", "g1_col", "g0_row", "g1_row"):
                state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, False, [], unexpected_keys, error_msgs)

    def forward(self, coeffs):
        yl, yh = coeffs
        for h in yh[::-1]:
            yl, _ = ops.idwt_haar(yl, h)
        return yl


class DWT(nn.Module):
    """`DWT(J, wave="haar", mode="reflect")`: forward(x) -> (yl, [yh_1 .. yh_J]).  For the 2-tap Haar filters every padding mode
    gives the same coefficients on even sizes; on an odd axis only mode="reflect" (the reference's, NYUv2/train.py:258) is
    implemented: one reflected sample on the right / bottom."""

    def __init__(self, J=1, wave="haar", mode="reflect"):
        super().__init__()
        if wave != "haar":
            raise NotImplementedError("only the Haar wavelet is used by the reference")
        self.J = J
        self.mode = mode

    def forward(self, x):
        if self.mode != "reflect":
            h, w = x.shape[-2:]
            for _ in range(self.J):
                if h % 2 or w % 2:
                    raise NotImplementedError("DWT on odd sizes is implemented for mode='reflect' only (got %r)" % self.mode)
                h, w = h // 2, w // 2
        return ops.dwt_haar(x, self.J)


DWTInverse = IDWT
DWTForward = DWT
