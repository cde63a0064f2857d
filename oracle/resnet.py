"""Oracle (test infrastructure): ResNet-50 v1.5 trunk without avgpool/fc, PyTorch-CPU fp32.

Restates ``pare.models.backbone.resnet50`` (un-vendored; a copy of torchvision's
``ResNet(Bottleneck, [3,4,6,3])`` returning the layer4 map).  Reference call sites:
``spec/models/hmr.py:53`` and ``camcalib/model.py:33``; the consumer applies its own
avg-pool to a 4-D map (``camcalib/model.py:74-75``) and expects 2048 channels (``:37``).
Stride sits on the 3x3 conv (v1.5).  State-dict keys are torchvision's.
"""
import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class ResNet50Trunk(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x, return_stages=False):
        stages = {}
        x = self.relu(self.bn1(self.conv1(x)))
        stages['stem'] = x
        x = self.maxpool(x)
        stages['pool'] = x
        x = self.layer1(x); stages['layer1'] = x
        x = self.layer2(x); stages['layer2'] = x
        x = self.layer3(x); stages['layer3'] = x
        x = self.layer4(x); stages['layer4'] = x
        return (x, stages) if return_stages else x


class BasicBlock(nn.Module):
    """torchvision BasicBlock (ResNet-18/34)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class ResNet34Trunk(ResNet50Trunk):
    """``pare.models.backbone.resnet34`` = torchvision ``ResNet(BasicBlock, [3,4,6,3])`` without avgpool / fc
    (camcalib/model.py:85, camcalib/config.py:81); 512 output channels."""

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)


def resnet34(pretrained=False, **kwargs):
    return ResNet34Trunk()


def resnet18(pretrained=False, **kwargs):
    """torchvision ``ResNet(BasicBlock, [2,2,2,2])`` (pare.models.backbone re-exports the torchvision family)."""
    return ResNet34Trunk(layers=(2, 2, 2, 2))


def resnet101(pretrained=False, **kwargs):
    return ResNet50Trunk(layers=(3, 4, 23, 3))


def resnet152(pretrained=False, **kwargs):
    return ResNet50Trunk(layers=(3, 8, 36, 3))


def resnet50(pretrained=False, **kwargs):
    """``pretrained`` is accepted and ignored: no network, weights come from a state_dict."""
    return ResNet50Trunk()


def get_backbone_info(backbone):
    """``pare.models.backbone.utils.get_backbone_info`` (call sites ``spec/models/hmr.py:58``,
    ``camcalib/model.py:37``)."""
    info = {
        'resnet18': {'n_output_channels': 512}, 'resnet34': {'n_output_channels': 512},
        'resnet50': {'n_output_channels': 2048}, 'resnet101': {'n_output_channels': 2048},
        'resnet152': {'n_output_channels': 2048},
        'hrnet_w32': {'n_output_channels': 480}, 'hrnet_w48': {'n_output_channels': 720},
    }
    return info[backbone]
