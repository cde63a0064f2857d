"""Oracle (test infrastructure): HRNet-W32 / W48 trunk as HMR builds it - ``eval(backbone)(pretrained=True,
downsample=True, use_conv=...)`` for ``hrnet_w32-conv`` / ``-interp`` (spec/models/hmr.py:44-51) - restating the
un-vendored ``pare.models.backbone.hrnet`` (requirements.txt:28), i.e. ``PoseHighResolutionNet`` of the published
HRNet pose code (stem 2 x [3x3/s2 conv + BN + ReLU], ``layer1`` = 4 Bottlenecks, stages 2-4 of HighResolutionModules
with BASIC blocks [4,4(,4,4)] x NUM_MODULES (1, 4, 3), SUM fusion with nearest upsampling) plus PARE's multi-scale head
(``downsample=True``: strided-conv stacks ``downsample_stage_{1,2,3}`` or bilinear ``align_corners=True`` interpolation,
then channel concatenation: 480 / 720 channels).  Parity unpinned (no upstream source or weights in this image):
pinned only by structure (state-dict key list, output shape / ``n_output_channels``) and closed-form tests.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_MOMENTUM = 0.1


def conv3x3(i, o, stride=1):
    return nn.Conv2d(i, o, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu(out + residual)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        return self.relu(out + residual)


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, num_blocks, num_channels):
        super().__init__()
        self.num_branches = num_branches
        self.num_inchannels = list(num_channels)
        self.branches = nn.ModuleList([nn.Sequential(*[BasicBlock(num_channels[b], num_channels[b]) for _ in range(num_blocks[b])])
                                       for b in range(num_branches)])
        self.fuse_layers = self._make_fuse_layers()
        self.relu = nn.ReLU(True)

    def _make_fuse_layers(self):
        nb, c = self.num_branches, self.num_inchannels
        fuse_layers = []
        for i in range(nb):
            fuse_layer = []
            for j in range(nb):
                if j > i:
                    fuse_layer.append(nn.Sequential(nn.Conv2d(c[j], c[i], 1, 1, 0, bias=False), nn.BatchNorm2d(c[i]),
                                                    nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    fuse_layer.append(None)
                else:
                    convs = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            convs.append(nn.Sequential(nn.Conv2d(c[j], c[i], 3, 2, 1, bias=False), nn.BatchNorm2d(c[i])))
                        else:
                            convs.append(nn.Sequential(nn.Conv2d(c[j], c[j], 3, 2, 1, bias=False), nn.BatchNorm2d(c[j]),
                                                       nn.ReLU(True)))
                    fuse_layer.append(nn.Sequential(*convs))
            fuse_layers.append(nn.ModuleList(fuse_layer))
        return nn.ModuleList(fuse_layers)

    def forward(self, x):
        for i in range(self.num_branches):
            x[i] = self.branches[i](x[i])
        x_fuse = []
        for i in range(len(self.fuse_layers)):
            y = x[0] if i == 0 else self.fuse_layers[i][0](x[0])
            for j in range(1, self.num_branches):
                if i == j:
                    y = y + x[j]
                else:
                    y = y + self.fuse_layers[i][j](x[j])
            x_fuse.append(self.relu(y))
        return x_fuse


class PoseHighResolutionNet(nn.Module):
    def __init__(self, width=32, downsample=True, use_conv=True):
        super().__init__()
        assert downsample, 'HMR builds the HRNet trunks with downsample=True (spec/models/hmr.py:47-50)'
        C = [width, width * 2, width * 4, width * 8]
        self.width, self.use_conv = width, use_conv
        self.conv1 = nn.Conv2d(3, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=True)
        ds = nn.Sequential(nn.Conv2d(64, 256, kernel_size=1, stride=1, bias=False), nn.BatchNorm2d(256, momentum=BN_MOMENTUM))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, 1, ds), Bottleneck(256, 64), Bottleneck(256, 64), Bottleneck(256, 64))
        self.transition1 = nn.ModuleList([
            nn.Sequential(nn.Conv2d(256, C[0], 3, 1, 1, bias=False), nn.BatchNorm2d(C[0]), nn.ReLU(inplace=True)),
            nn.Sequential(nn.Sequential(nn.Conv2d(256, C[1], 3, 2, 1, bias=False), nn.BatchNorm2d(C[1]), nn.ReLU(inplace=True)))])
        self.stage2 = nn.Sequential(*[HighResolutionModule(2, [4, 4], C[:2]) for _ in range(1)])
        self.transition2 = nn.ModuleList([None, None, nn.Sequential(nn.Sequential(
            nn.Conv2d(C[1], C[2], 3, 2, 1, bias=False), nn.BatchNorm2d(C[2]), nn.ReLU(inplace=True)))])
        self.stage3 = nn.Sequential(*[HighResolutionModule(3, [4, 4, 4], C[:3]) for _ in range(4)])
        self.transition3 = nn.ModuleList([None, None, None, nn.Sequential(nn.Sequential(
            nn.Conv2d(C[2], C[3], 3, 2, 1, bias=False), nn.BatchNorm2d(C[3]), nn.ReLU(inplace=True)))])
        self.stage4 = nn.Sequential(*[HighResolutionModule(4, [4, 4, 4, 4], C) for _ in range(3)])
        if use_conv:
            self.downsample_stage_1 = self._make_downsample_layer(3, C[0])
            self.downsample_stage_2 = self._make_downsample_layer(2, C[1])
            self.downsample_stage_3 = self._make_downsample_layer(1, C[2])

    @staticmethod
    def _make_downsample_layer(num_layers, num_channel, kernel_size=3):
        layers = []
        for _ in range(num_layers):
            layers += [nn.Conv2d(num_channel, num_channel, kernel_size=kernel_size, stride=2, padding=1, bias=False),
                       nn.BatchNorm2d(num_channel, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        x_list = [self.transition1[0](x), self.transition1[1](x)]
        y_list = self.stage2(x_list)
        x_list = [y_list[0], y_list[1], self.transition2[2](y_list[-1])]
        y_list = self.stage3(x_list)
        x_list = [y_list[0], y_list[1], y_list[2], self.transition3[3](y_list[-1])]
        x = self.stage4(x_list)
        if self.use_conv:
            x1, x2, x3 = self.downsample_stage_1(x[0]), self.downsample_stage_2(x[1]), self.downsample_stage_3(x[2])
        else:
            size = (x[3].size(2), x[3].size(3))
            x1, x2, x3 = (F.interpolate(t, size=size, mode='bilinear', align_corners=True) for t in x[:3])
        return torch.cat([x1, x2, x3, x[3]], 1)


def hrnet_w32(pretrained=False, downsample=True, use_conv=True, **kw):
    return PoseHighResolutionNet(32, downsample, use_conv)


def hrnet_w48(pretrained=False, downsample=True, use_conv=True, **kw):
    return PoseHighResolutionNet(48, downsample, use_conv)
