"""Layer modules with the reference's names, constructor arguments and state_dict keys, executing on
libwmd_hip.so.  Parameters live in plain nn.Conv2d holders so that checkpoints and the KITTI
trainer's `group_weight` parameter-group splitter (KITTI/pyt_utils.py:12-28, which asserts every
parameter belongs to a Conv2d/Linear) keep working; the holders' own forward is never used.

  KITTI flavour: Conv3x3(in, out, use_refl=True), Conv1x1, ConvBlock(in, out, kernel_size, use_refl)
                 (KITTI/layers.py:120-173)
  NYUv2 flavour: NyuConv3x3(in, out, padding=...), UpSampleBlock (NYUv2/networks/layers.py:11-32,57-67)
"""
import torch.nn as nn

from . import ops


class DeferredActivation:
    """An encoder's last feature map handed over BEFORE its final activation (the encoder edge):
    `tensor` is the pre-activation, the consumer reads it as  act(tensor * scale[c] + shift[c])  (scale / shift optional
    per-channel float32 tensors: an eval-mode BatchNorm folded into the edge; act in {"none", "leaky"}, ReLU = leaky with
    slope 0).  The decoders hand it to ops.conv2d_pre_activated, which activates it in one elementwise pass and runs the tuned
    convolution (round 3's activation-on-load kernels measured slower and are gone); under autograd they call `activate()` and
    run the ordinary path.  encoders.ResnetEncoder(defer_last_relu=True) produces one; off by default: an interface for
    encoders that end before their activation, not a speed-up."""

    def __init__(self, tensor, act="leaky", slope=0.0, scale=None, shift=None):
        if act not in ("none", "leaky"):
            raise ValueError("DeferredActivation: act must be 'none' or 'leaky' (ReLU = leaky with slope 0)")
        self.tensor, self.act, self.slope, self.scale, self.shift = tensor, act, float(slope), scale, shift

    @property
    def shape(self):
        return self.tensor.shape

    def pre(self):
        return (self.scale, self.shift, self.act, self.slope)

    def key(self):
        return (self.act, self.slope, None if self.scale is None else self.scale.data_ptr(),
                None if self.shift is None else self.shift.data_ptr())

    def activate(self):
        v = self.tensor
        if self.scale is not None:
            v = v * self.scale.view(1, -1, 1, 1)
        if self.shift is not None:
            v = v + self.shift.view(1, -1, 1, 1)
        return nn.functional.leaky_relu(v, self.slope) if self.act == "leaky" else v


def split_edge(input_features):
    """-> (list of plain tensors, DeferredActivation or None) for a feature list whose last entry may be deferred."""
    feats = list(input_features)
    edge = feats[-1] if isinstance(feats[-1], DeferredActivation) else None
    if edge is not None:
        feats[-1] = edge.tensor
    return feats, edge


def gated_backward_allowed(decoder):
    """The decoders fold every trunk activation's derivative into the CONSUMERS' data gradients (`x1_gate`) and tell the
    producer that what arrives is already the pre-activation gradient (`grad_is_dz`).  That is only sound while the decoder's
    own call graph is the sole consumer of those activations: a forward / backward hook on any sub-module (or a global
    module hook) can hand an activation to somebody who does not apply the gate, whose gradient would then silently lack
    ELU' / LeakyReLU'.  With a hook registered -- or WMD_GATED_BWD=0 -- the hints are dropped and every convolution runs its
    own activation-backward pass (general, ~0.2-0.3 ms slower per training step)."""
    import os
    import torch.nn.modules.module as tm
    if os.environ.get("WMD_GATED_BWD", "1") == "0":
        return False
    for name in ("_global_forward_hooks", "_global_forward_pre_hooks", "_global_backward_hooks", "_global_backward_pre_hooks"):
        if getattr(tm, name, None):
            return False
    for m in decoder.modules():
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, "_backward_pre_hooks", None):
            return False
    return True


class Conv3x3(nn.Module):
    """Pad (reflect or zero) and convolve — KITTI/layers.py:146-161."""

    def __init__(self, in_channels, out_channels, use_refl=True, stride=1, use_bias=True):
        super().__init__()
        if stride != 1:
            raise NotImplementedError("the decoders only use stride 1")
        self.pad_mode = "reflect" if use_refl else "zero"
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3, stride=stride, bias=use_bias)

    def forward(self, x, skip=None, up=1, act="none", slope=0.0, x1_gate=None, grad_is_dz=False, x1_pre=None):
        if x1_pre is not None:      # encoder edge (inference): x is a pre-activation, activated on load
            assert skip is None
            return ops.conv2d_pre_activated(x, x1_pre, self.conv.weight, self.conv.bias, up1=up, pad=self.pad_mode, act=act, slope=slope)
        return ops.conv2d_fused(x, self.conv.weight, self.conv.bias, x2=skip, up1=up, pad=self.pad_mode, act=act,
                                slope=slope, x1_gate=x1_gate, grad_is_dz=grad_is_dz)


class Conv1x1(nn.Module):
    """KITTI/layers.py:164-173."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 1, stride=1, padding=0)

    def forward(self, x, act="none", slope=0.0, x1_gate=None, grad_is_dz=False):
        return ops.conv2d_fused(x, self.conv.weight, self.conv.bias, pad="zero", act=act, slope=slope, x1_gate=x1_gate,
                                grad_is_dz=grad_is_dz)


class ConvBlock(nn.Module):
    """Convolution followed by ELU — KITTI/layers.py:120-143 (norm_layer is Identity everywhere)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, norm_layer=None, use_refl=False):
        super().__init__()
        if norm_layer is not None:
            raise NotImplementedError("no decoder of the reference passes a norm_layer")
        if kernel_size == 3:
            self.conv = Conv3x3(in_channels, out_channels, use_refl=use_refl)
        elif kernel_size == 1:
            self.conv = Conv1x1(in_channels, out_channels)
        else:
            raise NotImplementedError
        self.kernel_size = kernel_size

    def forward(self, x, skip=None, up=1, x1_gate=None, grad_is_dz=False, x1_pre=None):
        if self.kernel_size == 3:
            return self.conv(x, skip=skip, up=up, act="elu", x1_gate=x1_gate, grad_is_dz=grad_is_dz, x1_pre=x1_pre)
        if x1_pre is not None:
            raise NotImplementedError("the encoder edge feeds a 3x3 ConvBlock")
        return self.conv(x, act="elu", x1_gate=x1_gate, grad_is_dz=grad_is_dz)


class NyuConv3x3(nn.Module):
    """NYUv2/networks/layers.py:11-32 (`padding` in {"reflection","replicate","zero"}).  With is_depthwise the layer is
    depthwise 3x3 (no bias) -> ReLU -> 1x1 (no bias) (:23-25,70-79; state_dict keys conv.0.0.weight, conv.1.weight)."""

    def __init__(self, in_channels, out_channels, padding="zero", stride=1, is_depthwise=False):
        super().__init__()
        if stride != 1:
            raise NotImplementedError
        self.pad_mode = {"reflection": "reflect", "replicate": "replicate"}.get(padding, "zero")
        self.is_depthwise = bool(is_depthwise)
        if self.is_depthwise:
            dw = nn.Sequential(nn.Conv2d(int(in_channels), int(in_channels), 3, stride=1, padding=0, bias=False, groups=int(in_channels)),
                               nn.ReLU(inplace=True))
            self.conv = nn.Sequential(dw, nn.Conv2d(int(in_channels), int(out_channels), 1, 1, 0, bias=False))
        else:
            self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3, stride=stride, padding=0)

    def forward(self, x, skip=None, up=1, act="none", slope=0.0, x1_gate=None, grad_is_dz=False):
        # x1_gate / grad_is_dz: backward-only hints of ops.conv2d_fused (plain 3x3 layers only)
        if self.is_depthwise:
            mid = ops.dwconv3x3_relu(x, self.conv[0][0].weight, x2=skip, up1=up, pad=self.pad_mode)
            return ops.conv2d_fused(mid, self.conv[1].weight, None, pad="zero", act=act, slope=slope)
        return ops.conv2d_fused(x, self.conv.weight, self.conv.bias, x2=skip, up1=up, pad=self.pad_mode, act=act,
                                slope=slope, x1_gate=x1_gate, grad_is_dz=grad_is_dz)

    def head(self, x, scale, x_gate=None):
        """scale * layer(x) for the 1- and 3-channel wavelet heads."""
        if self.is_depthwise:
            return self.forward(x) * scale
        return ops.head3x3(x, self.conv.weight, self.conv.bias, pad=self.pad_mode, mode=0, scale=scale, x_gate=x_gate)


class UpSampleBlock(nn.Module):
    """nearest x2 -> cat(skip) -> Conv3x3 -> LeakyReLU(0.2) — NYUv2/networks/layers.py:57-67,
    executed as one fused kernel."""

    def __init__(self, skip_input, output_features, padding="zero", is_depthwise=False):
        super().__init__()
        self.convA = NyuConv3x3(skip_input, output_features, padding=padding, is_depthwise=is_depthwise)

    def forward(self, x, concat_with, x1_gate=None, grad_is_dz=False):
        return self.convA(x, skip=concat_with, up=2, act="leaky", slope=0.2, x1_gate=x1_gate, grad_is_dz=grad_is_dz)
