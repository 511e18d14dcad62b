"""Plain torch.nn ResNet encoder (stays PyTorch-ROCm by design: BASELINE.json north_star keeps the encoder as
ordinary modules).  torchvision is not installed in this image, so the network is defined here with
torchvision-compatible state_dict names (`encoder.conv1.weight`, `encoder.layer1.0.conv1.weight`, ...), which is
what the reference's `encoder.pth` checkpoints contain (KITTI/networks/encoders/resnet_encoder.py:62-98).
Returns the five feature maps (strides 2..32) the decoders consume; `num_ch_enc` as in the reference (:68,84-85)."""
import numpy as np
import torch
import torch.nn as nn


class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, down=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = down

    def forward(self, x, defer_relu=False):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return out + idt if defer_relu else self.relu(out + idt)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, down=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = down

    def forward(self, x, defer_relu=False):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return out + idt if defer_relu else self.relu(out + idt)


class _ResNet(nn.Module):
    def __init__(self, block, layers, num_input_images=1):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0])
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, 1000)   # unused by the depth network (grad stays None)

    def _make(self, block, planes, n, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*layers)


_SPECS = {18: (_Basic, [2, 2, 2, 2]), 34: (_Basic, [3, 4, 6, 3]), 50: (_Bottleneck, [3, 4, 6, 3]),
          101: (_Bottleneck, [3, 4, 23, 3]), 152: (_Bottleneck, [3, 8, 36, 3])}


class ResnetEncoder(nn.Module):
    """defer_last_relu (extension): in inference the last residual block returns `out + identity` WITHOUT
    its final ReLU, wrapped as layers.DeferredActivation -- the decoders' first convolution applies the ReLU while it loads the
    tensor, so the activated stride-32 map is neither written by the encoder nor re-read.  Under autograd the ordinary
    activated tensor is returned."""

    def __init__(self, num_layers=18, pretrained=False, num_input_images=1, defer_last_relu=False):
        super().__init__()
        self.defer_last_relu = bool(defer_last_relu)
        if pretrained:
            raise RuntimeError("no network access in this environment: load encoder.pth explicitly")
        block, layers = _SPECS[num_layers]
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self.encoder = _ResNet(block, layers, num_input_images)

    def forward(self, input_image):
        e = self.encoder
        x = (input_image - 0.45) / 0.225
        f0 = e.relu(e.bn1(e.conv1(x)))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        if self.defer_last_relu and not torch.is_grad_enabled():
            from .layers import DeferredActivation
            h = f3
            for blk in list(e.layer4)[:-1]:
                h = blk(h)
            return [f0, f1, f2, f3, DeferredActivation(e.layer4[-1](h, defer_relu=True), act="leaky", slope=0.0)]
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]


# ---------------------------------------------------------------------------------------------
# DenseNet (NYUv2 config 5) and MobileNetV2: also ordinary PyTorch, torchvision-compatible names
# ---------------------------------------------------------------------------------------------

class _DenseLayer(nn.Module):
    """BN-ReLU-1x1(bn_size*growth) -> BN-ReLU-3x3(growth); consumes the concatenation of everything before it."""

    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, x):
        return self.conv2(self.relu2(self.norm2(self.conv1(self.relu1(self.norm1(x))))))


class _DenseBlock(nn.Module):
    def __init__(self, n, cin, growth, bn_size):
        super().__init__()
        self.cin, self.growth, self.n = cin, growth, n
        for k in range(n):
            self.add_module("denselayer%d" % (k + 1), _DenseLayer(cin + k * growth, growth, bn_size))

    def forward(self, x):
        if torch.is_grad_enabled():
            # training: torchvision's form (autograd cannot track slice writes into a buffer earlier layers still read)
            feats = [x]
            for k in range(self.n):
                feats.append(getattr(self, "denselayer%d" % (k + 1))(torch.cat(feats, 1)))
            return torch.cat(feats, 1)
        # inference: one preallocated [B, cin + n*growth, H, W] slab; layer k reads a channel prefix and writes its
        # slice -- same values, without the O(n^2) concatenation copies
        B, _, H, W = x.shape
        slab = x.new_empty((B, self.cin + self.n * self.growth, H, W))
        slab[:, :self.cin] = x
        c = self.cin
        for k in range(self.n):
            slab[:, c:c + self.growth] = getattr(self, "denselayer%d" % (k + 1))(slab[:, :c])
            c += self.growth
        return slab


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("norm", nn.BatchNorm2d(cin))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", nn.Conv2d(cin, cout, 1, bias=False))
        self.add_module("pool", nn.AvgPool2d(2, 2))


class _DenseNet(nn.Module):
    """`features` Sequential with torchvision's child names: conv0 norm0 relu0 pool0 denseblock1 transition1 ...
    denseblock4 norm5, plus the (unused) classifier so an ImageNet checkpoint loads with strict=True."""

    def __init__(self, growth, blocks, init_features, bn_size=4):
        super().__init__()
        f = nn.Sequential()
        f.add_module("conv0", nn.Conv2d(3, init_features, 7, 2, 3, bias=False))
        f.add_module("norm0", nn.BatchNorm2d(init_features))
        f.add_module("relu0", nn.ReLU(inplace=True))
        f.add_module("pool0", nn.MaxPool2d(3, 2, 1))
        c = init_features
        for k, n in enumerate(blocks):
            f.add_module("denseblock%d" % (k + 1), _DenseBlock(n, c, growth, bn_size))
            c += n * growth
            if k != len(blocks) - 1:
                f.add_module("transition%d" % (k + 1), _Transition(c, c // 2))
                c //= 2
        f.add_module("norm5", nn.BatchNorm2d(c))
        self.features = f
        self.classifier = nn.Linear(c, 1000)


_DENSE_SPECS = {121: (32, (6, 12, 24, 16), 64), 161: (48, (6, 12, 36, 24), 96), 169: (32, (6, 12, 32, 32), 64),
                201: (32, (6, 12, 48, 32), 64)}


class DenseEncoder(nn.Module):
    """NYUv2/networks/encoders/densenet_encoder.py:4-33.  The reference always instantiates densenet161 whatever
    `num_layers` says (:17) and returns the outputs of relu0, pool0, transition1, transition2 and denseblock4
    (features 3, 4, 6, 8, 11 of its running list), i.e. `num_ch_enc = [96, 96, 192, 384, 2208]`; its input normalisation
    loop discards its result (:26-28), so `normalize_input` changes nothing there or here.  norm5 is not evaluated (the
    reference computes and drops it)."""

    def __init__(self, normalize_input=True, num_layers=161, pretrained=False):
        super().__init__()
        if num_layers not in _DENSE_SPECS:
            raise AssertionError("Can't use any number of layers, should use from 121, 161, 169, 201")
        if pretrained:
            raise RuntimeError("no network access in this environment: load the encoder weights explicitly")
        self.original_model = _DenseNet(*_DENSE_SPECS[161])
        self.normalize_input = normalize_input
        self.num_ch_enc = [96, 96, 192, 384, 2208]

    def forward(self, x):
        f = self.original_model.features
        f0 = f.relu0(f.norm0(f.conv0(x)))
        f1 = f.pool0(f0)
        f2 = f.transition1(f.denseblock1(f1))
        f3 = f.transition2(f.denseblock2(f2))
        f4 = f.denseblock4(f.transition3(f.denseblock3(f3)))
        return f0, f1, f2, f3, f4


class _ComposedReLU6(nn.Module):
    """`use_custom_relu6` of the reference (mobilenetv2_encoder.py:18-30, its ONNX-export form): min(relu(x), 6) written as
    6 - relu(6 - relu(x)); no parameters, no buffers (the state_dict is the same either way)."""

    def forward(self, x):
        return 6.0 - torch.relu(6.0 - torch.relu(x))


def _cbr6(cin, cout, k=3, stride=1, groups=1, custom=False):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False), nn.BatchNorm2d(cout),
                         _ComposedReLU6() if custom else nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, expand, custom=False):
        super().__init__()
        hid = int(round(cin * expand))
        self.skip = stride == 1 and cin == cout
        seq = [] if expand == 1 else [_cbr6(cin, hid, 1, custom=custom)]
        seq += [_cbr6(hid, hid, 3, stride, groups=hid, custom=custom), nn.Conv2d(hid, cout, 1, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*seq)

    def forward(self, x):
        y = self.conv(x)
        return x + y if self.skip else y


class MobileNetV2Encoder(nn.Module):
    """KITTI|NYUv2/networks/encoders/mobilenetv2_encoder.py: `features.N...` names of torchvision's mobilenet_v2;
    the five maps are the stem and the first block of every stride-2 stage, the last one replaced by the 1x1 -> 1280 layer
    when `use_last_layer`; `num_ch_enc` = [32, 24, 32, 64, 1280] or [..., 160]."""
    _STAGES = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2)]

    def __init__(self, pretrained=False, use_custom_relu6=False, width_mult=1., use_last_layer=True, normalize_input=False,
                 num_layers=1):
        super().__init__()
        if pretrained:
            raise RuntimeError("no network access in this environment: load the encoder weights explicitly")
        self.use_last_layer = use_last_layer
        self.normalize_input = normalize_input   # the reference's normalisation loop has no effect (result discarded)
        c = int(32 * width_mult)
        chans, taps = [c], []
        feats = [_cbr6(3, c, 3, 2, custom=use_custom_relu6)]
        for t_, co, n, s in self._STAGES:
            co = int(co * width_mult)
            for r in range(n):
                feats.append(_InvertedResidual(c, co, s if r == 0 else 1, t_, custom=use_custom_relu6))
                c = co
                if s == 2 and r == 0:
                    chans.append(co)
                    taps.append(len(feats) - 1)
        if use_last_layer:
            feats.append(_cbr6(c, 1280, 1, custom=use_custom_relu6))
            chans[-1] = 1280
        self.features = nn.ModuleList(feats)
        self._taps = set(taps)
        self.num_ch_enc = np.asarray(chans)
        self._initialize_weights()

    def _initialize_weights(self):
        """The reference's starting point for training from scratch (mobilenetv2_encoder.py:136,160-173): convolution weights
        ~ N(0, sqrt(2 / (k * k * Cout))), BatchNorm weight 1 / bias 0 (torch's own Conv2d default is Kaiming-uniform)."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, 0.0, (2.0 / n) ** 0.5)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, input_image):
        x = self.features[0](input_image)
        outs = [x]
        n_blocks = len(self.features) - (1 if self.use_last_layer else 0)
        for k in range(1, n_blocks):
            x = self.features[k](x)
            if k in self._taps:
                outs.append(x)
        if self.use_last_layer:
            outs[-1] = self.features[-1](x)
        self.encoder_features = outs
        return outs
