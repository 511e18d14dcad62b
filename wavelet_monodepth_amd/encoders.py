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

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


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

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


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
    def __init__(self, num_layers=18, pretrained=False, num_input_images=1):
        super().__init__()
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
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]
