"""Drop-in ``nn.Module`` front ends of the SPEC hot path on MI355X.

``HMR`` mirrors ``spec/models/hmr.py:28-122`` and ``CameraRegressorNetwork`` mirrors
``camcalib/model.py:24-81``: same constructor signatures, forward signatures (positional and
keyword), output containers/keys and ``state_dict`` key layout, so the
reference's checkpoints and its callers (``spec/tester.py:53-59,143-151``,
``spec/trainer.py:50-56,138-139``, ``scripts/camcalib_demo.py:74-81,102``) work unchanged.

The sub-modules below only HOLD parameters under the reference's names (they are real
``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.Linear`` containers so ``load_state_dict`` /
``.to()`` / ``state_dict()`` behave), their torch forward is never executed.  ``forward`` hands
device pointers to libspecmi (hand-written HIP, gfx950).  There is no CPU path: a CPU input
raises.  Inference only (eval + no_grad semantics: BatchNorm uses running statistics and
dropout is the identity, as in ``spec/tester.py:47,90``).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import assets
from .engine import Engine


# --------------------------------------------------------------------------------------------
# parameter containers (names == reference state_dict keys)
# --------------------------------------------------------------------------------------------
class _BottleneckParams(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))


class ResNet50Params(nn.Module):
    """torchvision-layout Bottleneck ResNet trunk parameters (no avgpool / fc): ResNet-50 by default (318 tensors), ``layers``
    = (3, 4, 23, 3) / (3, 8, 36, 3) for ResNet-101 / 152."""

    def __init__(self, layers=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (nb, planes) in enumerate(zip(layers, (64, 128, 256, 512)), start=1):
            blocks = []
            for b in range(nb):
                stride = 2 if (b == 0 and li > 1) else 1
                blocks.append(_BottleneckParams(inplanes, planes, stride, downsample=(b == 0)))
                inplanes = planes * 4
            setattr(self, f'layer{li}', nn.Sequential(*blocks))

    def forward(self, *a, **k):
        raise RuntimeError('ResNet50Params only holds parameters; the trunk runs inside libspecmi')


class _BasicBlockParams(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes))


class ResNet34Params(nn.Module):
    """torchvision-layout BasicBlock ResNet trunk parameters (no avgpool / fc): ResNet-34 ([3,4,6,3]) - the CamCalib
    config default (camcalib/config.py:81) and one of the two trunks of camcalib/model.py:85 - or ResNet-18 ([2,2,2,2])."""

    def __init__(self, layers=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (nb, planes) in enumerate(zip(layers, (64, 128, 256, 512)), start=1):
            blocks = []
            for b in range(nb):
                stride = 2 if (b == 0 and li > 1) else 1
                blocks.append(_BasicBlockParams(inplanes, planes, stride, downsample=(stride != 1 or inplanes != planes)))
                inplanes = planes
            setattr(self, f'layer{li}', nn.Sequential(*blocks))

    def forward(self, *a, **k):
        raise RuntimeError('ResNet34Params only holds parameters; the trunk runs inside libspecmi')


def resnet34(pretrained=False, **kwargs):
    return ResNet34Params()


def resnet18(pretrained=False, **kwargs):
    return ResNet34Params((2, 2, 2, 2))


def resnet101(pretrained=False, **kwargs):
    return ResNet50Params((3, 4, 23, 3))


def resnet152(pretrained=False, **kwargs):
    return ResNet50Params((3, 8, 36, 3))


# ---- HRNet-W32 / W48 (pare.models.backbone.hrnet as HMR builds it, spec/models/hmr.py:44-51) -------------------
def _cbn(cin, cout, k, stride, relu=False, extra=None):
    """Sequential(conv, bn[, relu][, extra]) - index layout of the upstream state_dict ('.0' conv, '.1' bn)."""
    mods = [nn.Conv2d(cin, cout, k, stride, 1 if k == 3 else 0, bias=False), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    if extra is not None:
        mods.append(extra)
    return nn.Sequential(*mods)


class _HRModuleParams(nn.Module):
    """HighResolutionModule: ``branches.{b}.{k}.(conv1|bn1|conv2|bn2)`` and ``fuse_layers.{i}.{j}...``."""

    def __init__(self, channels):
        super().__init__()
        nb = len(channels)
        self.branches = nn.ModuleList([nn.Sequential(*[_BasicBlockParams(c, c, 1, False) for _ in range(4)]) for c in channels])
        fuse = []
        for i in range(nb):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(_cbn(channels[j], channels[i], 1, 1, extra=nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(None)
                else:
                    row.append(nn.Sequential(*[_cbn(channels[j], channels[i] if k == i - j - 1 else channels[j], 3, 2,
                                                    relu=(k != i - j - 1)) for k in range(i - j)]))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse)


class HRNetParams(nn.Module):
    """Parameter container with the upstream PoseHighResolutionNet key layout (downsample=True head)."""

    def __init__(self, width=32, use_conv=True):
        super().__init__()
        C = [width, width * 2, width * 4, width * 8]
        self.width, self.use_conv = width, use_conv
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        self.layer1 = nn.Sequential(*[_BottleneckParams(64 if b == 0 else 256, 64, 1, downsample=(b == 0)) for b in range(4)])
        self.transition1 = nn.ModuleList([_cbn(256, C[0], 3, 1, relu=True), nn.Sequential(_cbn(256, C[1], 3, 2, relu=True))])
        self.stage2 = nn.Sequential(_HRModuleParams(C[:2]))
        self.transition2 = nn.ModuleList([None, None, nn.Sequential(_cbn(C[1], C[2], 3, 2, relu=True))])
        self.stage3 = nn.Sequential(*[_HRModuleParams(C[:3]) for _ in range(4)])
        self.transition3 = nn.ModuleList([None, None, None, nn.Sequential(_cbn(C[2], C[3], 3, 2, relu=True))])
        self.stage4 = nn.Sequential(*[_HRModuleParams(C) for _ in range(3)])
        if use_conv:
            for d, n in enumerate((3, 2, 1)):
                mods = []
                for _ in range(n):
                    mods += [nn.Conv2d(C[d], C[d], 3, 2, 1, bias=False), nn.BatchNorm2d(C[d]), nn.ReLU(inplace=True)]
                setattr(self, f'downsample_stage_{d + 1}', nn.Sequential(*mods))

    def forward(self, *a, **k):
        raise RuntimeError('HRNetParams only holds parameters; the trunk runs inside libspecmi')


def hrnet_w32(pretrained=False, downsample=True, use_conv=True, **kwargs):
    if not downsample:
        raise NotImplementedError('HMR builds the HRNet trunks with downsample=True (spec/models/hmr.py:47-50)')
    return HRNetParams(32, use_conv)


def hrnet_w48(pretrained=False, downsample=True, use_conv=True, **kwargs):
    if not downsample:
        raise NotImplementedError('HMR builds the HRNet trunks with downsample=True (spec/models/hmr.py:47-50)')
    return HRNetParams(48, use_conv)


def resnet50(pretrained=False, **kwargs):
    """Name looked up by the reference via ``eval(backbone)`` (hmr.py:53, camcalib/model.py:33).
    ``pretrained`` never touches the network here; weights come from a checkpoint."""
    return ResNet50Params()


def get_backbone_info(backbone):
    """pare.models.backbone.utils.get_backbone_info: the trunk's feature width (HRNet, downsample=True head: the four
    branches concatenated, 32+64+128+256 / 48+96+192+384)."""
    return {'resnet50': {'n_output_channels': 2048}, 'resnet34': {'n_output_channels': 512},
            'resnet18': {'n_output_channels': 512}, 'resnet101': {'n_output_channels': 2048},
            'resnet152': {'n_output_channels': 2048},
            'hrnet_w32': {'n_output_channels': 480}, 'hrnet_w48': {'n_output_channels': 720}}[backbone]


# the torchvision ResNet family the reference's ``eval(backbone)(pretrained=True)`` resolves through pare.models.backbone
# (spec/models/hmr.py:53, camcalib/model.py:33): name -> (constructor, depth id of the library's "backbone" option)
def _resnet_family():
    return {'resnet18': (resnet18, 18), 'resnet34': (resnet34, 34), 'resnet50': (resnet50, 50), 'resnet101': (resnet101, 101),
            'resnet152': (resnet152, 152)}


UNCERTAINTY_ACTIVATIONS = {'': 0, 'relu': 1, 'softplus': 2, 'sigmoid': 3, 'tanh': 4, 'elu': 5}     # F.<name> the reference evaluates


class HMRHeadParams(nn.Module):
    """Parameters of pare's HMRHead in its state_dict layout (call site spec/models/hmr.py:57-64).  ``estimate_var``: the decoders
    also emit a variance per pose / shape number - doubled ``decpose`` (288 rows: mean | var) / ``decshape`` (20), or, with
    ``use_separate_var_branch``, the plain decoders plus ``decpose_var`` (144) / ``decshape_var`` (10)."""

    def __init__(self, num_input_features=2048, use_cam_feats=False, mean_params=None, estimate_var=False,
                 use_separate_var_branch=False):
        super().__init__()
        npose = 24 * 6
        nin = num_input_features + (7 if use_cam_feats else 0) + npose + 13
        self.fc1 = nn.Linear(nin, 1024)
        self.fc2 = nn.Linear(1024, 1024)
        doubled = estimate_var and not use_separate_var_branch
        self.decpose = nn.Linear(1024, npose * (2 if doubled else 1))
        self.decshape = nn.Linear(1024, 10 * (2 if doubled else 1))
        self.deccam = nn.Linear(1024, 3)
        if estimate_var and use_separate_var_branch:
            self.decpose_var = nn.Linear(1024, npose)
            self.decshape_var = nn.Linear(1024, 10)
            nn.init.xavier_uniform_(self.decpose_var.weight, gain=0.01)
            nn.init.xavier_uniform_(self.decshape_var.weight, gain=0.01)
        nn.init.xavier_uniform_(self.decpose.weight, gain=0.01)
        nn.init.xavier_uniform_(self.decshape.weight, gain=0.01)
        nn.init.xavier_uniform_(self.deccam.weight, gain=0.01)
        mp = mean_params if mean_params is not None else assets.mean_params()
        f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(1, -1).copy())
        self.register_buffer('init_pose', f(mp['pose']))
        self.register_buffer('init_shape', f(mp['shape']))
        self.register_buffer('init_cam', f(mp['cam']))


class _SMPLBuffers(nn.Module):
    """Buffers of the body model under smplx's names (checkpoints carry them as smpl.smpl.*)."""

    def __init__(self, model):
        super().__init__()
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
        for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'J_regressor_extra'):
            self.register_buffer(k, t(model[k], torch.float32))
        self.register_buffer('parents', t(model['parents'], torch.long))
        self.register_buffer('extra_vertex_ids', t(model['extra_vertex_ids'], torch.long), persistent=False)
        self.register_buffer('joint_map', t(model['joint_map'], torch.long), persistent=False)

    def as_model(self):
        return {k: v.detach().cpu().numpy() for k, v in self._buffers.items()}


class SMPLHeadParams(nn.Module):
    def __init__(self, img_res=224, focal_length=5000.):
        super().__init__()
        self.smpl = _SMPLBuffers(assets.smpl_model())
        self.img_res = img_res
        self.focal_length = focal_length


# --------------------------------------------------------------------------------------------
# engine-backed base
# --------------------------------------------------------------------------------------------
class _EngineModule(nn.Module):
    _kind = ''

    def __init__(self):
        super().__init__()
        self._engine = None
        self._sig = None
        self._frozen = False
        self._tracked = None          # cached list of the parameter / buffer tensors (see _signature)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    def _invalidate(self):
        self._tracked = None
        self._sig = None

    def _apply(self, fn, *a, **k):    # .to() / .cuda() / .float() may replace the tensors
        self._invalidate()
        return super()._apply(fn, *a, **k)

    def __setattr__(self, name, value):
        # assigning a new nn.Parameter / buffer tensor / submodule replaces tensors the cached list still points at
        if isinstance(value, (torch.Tensor, nn.Module)) and '_tracked' in self.__dict__:
            self.__dict__['_tracked'] = None
            self.__dict__['_sig'] = None
        super().__setattr__(name, value)

    # Parameters are re-uploaded when any tensor was replaced or modified in place.  The tensor list is cached
    # (walking ~330 state_dict entries per forward matters at batch 1) and rebuilt after load_state_dict / _apply;
    # the per-forward check is (data_ptr, _version) of the cached tensors.  In-place edits through ``.data`` do not
    # bump ``_version``: call ``commit()`` after those.  Replaced tensors are caught: assignments on this module
    # invalidate the list (__setattr__), and for children (``m.backbone.conv1.weight = ...``, ``child.load_state_dict(...,
    # assign=True)``) the signature carries the identity of every parameter / buffer object as the module tree yields
    # them now - a walk over ~330 cached-attribute lookups, far cheaper than building a state_dict.
    def _signature(self):
        ids = tuple(id(p) for p in self.parameters()) + tuple(id(b) for b in self.buffers())
        if self._tracked is None or self.__dict__.get('_tracked_ids') != ids:
            self._tracked = list(self.state_dict(keep_vars=True).values())
            self.__dict__['_tracked_ids'] = ids
        return tuple((v.data_ptr(), v._version) for v in self._tracked)

    def _options(self):
        return {}

    def _engine_state(self, sd):
        """state_dict -> the tensors the library stages (identity unless a module's layout differs from the library's)."""
        return sd

    # Optional: the plain 1x1 / stride-1 convolutions of the ResNet trunk as sums of bf16 x bf16 piece products on the bf16
    # matrix cores (conv_bf16s.hip).  0 = exact fp32 MFMA (default, what the benchmark's headline measures), 6 = three-way
    # split, six products: fp32-class results; 3 = two-way split, three products: ~1e-5 on the trunk output.
    conv_precision = 0

    def set_conv_precision(self, terms: int):
        if terms not in (0, 3, 6):
            raise ValueError('conv_precision must be 0 (exact fp32), 3 or 6 (bf16 piece products)')
        self.conv_precision = int(terms)
        self._invalidate()
        if self._engine is not None:    # re-commit now (a frozen module skips the per-forward signature check)
            self.commit(self._engine.device, freeze=self._frozen)
        else:
            self._sig = ('stale',)
        return self

    # Execution plan of the trunk (include/specmi.h, option "plan"): 'throughput' = the kernels the batch-256 headline runs
    # (Winograd + 64x64 / 128x128 implicit GEMM); 'latency' = every convolution cut into K slices that fill the chip at batch
    # 1-8 (one canonical summation tree per layer); 'single' (round 5) = the latency plan with every 3x3 convolution on the sliced
    # direct kernel (no Winograd): what batch 1-2 wants (with the opt-in option "persist" the trunk behind the max-pool is then ONE
    # persistent launch - measured slower than the per-layer launches, default off);
    # 'auto' (default) = single up to 2 images per call, latency up to 10 (16 for a single trunk), throughput beyond.  Within a plan an
    # image's result is bit-identical whatever the batch size; between plans the last bits differ (contract: 1e-4).
    PLANS = {'auto': 0, 'throughput': 1, 'latency': 2, 'single': 3}
    plan = 'auto'

    def set_plan(self, plan: str):
        if plan not in self.PLANS:
            raise ValueError(f"plan must be one of {tuple(self.PLANS)}")
        self.plan = plan
        if self._engine is not None:
            self._engine.set_option('plan', self.PLANS[plan])    # read at every forward: no re-commit
        return self

    def _smpl_model(self):
        return None

    def commit(self, device=None, freeze=False):
        """(Re)build the packed HBM copy of the parameters.  ``freeze=True`` skips the
        per-forward change check afterwards (serving / benchmark loops)."""
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('spec_amd models run on the GPU only: move the module and inputs to "cuda"')
        if self._engine is None or self._engine.device != torch.device('cuda', device.index or 0):
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(self._kind, device)
        sd = {k: v for k, v in self.state_dict().items()
              if not k.startswith('smpl.') and v.dtype.is_floating_point}
        sd = self._engine_state(sd)
        self._engine.load(sd, smpl=self._smpl_model(), conv_precision=int(self.conv_precision), plan=self.PLANS[self.plan],
                          **self._options())
        self._tracked = None
        self._sig = self._signature()
        self._frozen = freeze
        return self

    def engine(self, device) -> Engine:
        if self._engine is None or not self._frozen and self._sig != self._signature():
            self.commit(device, freeze=self._frozen)
        return self._engine

    def train(self, mode=True):
        if mode:
            raise RuntimeError('spec_amd implements the inference path only (eval mode)')
        return super().train(False)


class CameraRegressorNetwork(_EngineModule):
    """camcalib/model.py:24-81.  forward(images) -> [vfov, pitch, roll] logits, each (B,256)."""
    _kind = 'camcalib'

    def __init__(self, backbone='resnet50', num_fc_layers=1, num_fc_channels=1024, num_out_channels=256):
        super().__init__()
        if backbone not in _resnet_family():
            raise NotImplementedError(f'backbone {backbone!r}: resnet50 (the released model), resnet18 / 34 / 101 / 152 are built')
        assert num_fc_layers > 0, 'Number of FC layers should be more than 0'
        if num_fc_layers > 3 or num_fc_channels > 1024 or num_fc_channels % 32:
            raise NotImplementedError('num_fc_layers <= 3 and num_fc_channels <= 1024 (multiple of 32) are built')
        ctor, self._backbone_depth = _resnet_family()[backbone]
        self.backbone = ctor(pretrained=True)
        self.num_fc_layers, self.num_fc_channels = num_fc_layers, num_fc_channels
        self.num_out_channels = num_out_channels
        out_channels = get_backbone_info(backbone)['n_output_channels']
        if num_fc_layers == 1:
            self.fc_vfov = nn.Linear(out_channels, num_out_channels)
            self.fc_pitch = nn.Linear(out_channels, num_out_channels)
            self.fc_roll = nn.Linear(out_channels, num_out_channels)
            for fc in (self.fc_vfov, self.fc_pitch, self.fc_roll):
                nn.init.normal_(fc.weight, mean=0, std=0.01)
                nn.init.constant_(fc.bias, 0)
        else:
            self.fc_vfov = self._get_fc_layers(num_fc_layers, num_fc_channels, out_channels)
            self.fc_pitch = self._get_fc_layers(num_fc_layers, num_fc_channels, out_channels)
            self.fc_roll = self._get_fc_layers(num_fc_layers, num_fc_channels, out_channels)
        super().train(False)

    def _get_fc_layers(self, num_layers, num_channels, inp_channels):
        """camcalib/model.py:59-70: Linear layers back to back, no activation in between."""
        modules = []
        for i in range(num_layers):
            if i == 0:
                modules.append(nn.Linear(inp_channels, num_channels))
            elif i == num_layers - 1:
                modules.append(nn.Linear(num_channels, self.num_out_channels))
            else:
                modules.append(nn.Linear(num_channels, num_channels))
        return nn.Sequential(*modules)

    def _options(self):
        return {'backbone': self._backbone_depth, 'num_fc_layers': int(self.num_fc_layers),
                'num_fc_channels': int(self.num_fc_channels)}

    @torch.no_grad()
    def forward(self, images):
        return self.engine(images.device).camcalib_forward(images)


class HMR(_EngineModule):
    """spec/models/hmr.py:28-122.  Output dict keys: smpl_vertices, smpl_joints3d, smpl_joints2d,
    pred_cam_t, pred_pose, pred_cam, pred_shape, pred_pose_6d."""
    _kind = 'hmr'

    def __init__(self, backbone='resnet50', focal_length=5000., img_res=224, pretrained=None, use_cam=False,
                 p=0.0, estimate_var=False, use_separate_var_branch=False, uncertainty_activation='',
                 use_cam_feats=False):
        super().__init__()
        if uncertainty_activation not in UNCERTAINTY_ACTIVATIONS:
            raise NotImplementedError(f'uncertainty_activation {uncertainty_activation!r}: one of {sorted(UNCERTAINTY_ACTIVATIONS)} '
                                      '(the reference evaluates torch.nn.functional.<name>)')
        self.estimate_var = bool(estimate_var)
        self.use_separate_var_branch = bool(use_separate_var_branch)
        self.uncertainty_activation = uncertainty_activation
        self._hrnet_use_conv = 1
        if backbone.startswith('hrnet'):                       # hrnet_w32-conv, hrnet_w32-interp (hmr.py:44-51)
            backbone, use_conv = backbone.split('-')
            if backbone not in ('hrnet_w32', 'hrnet_w48'):
                raise NotImplementedError(f'backbone {backbone!r}: hrnet_w32 and hrnet_w48 are built')
            self._hrnet_use_conv = int(use_conv == 'conv')
            self.backbone = (hrnet_w32 if backbone == 'hrnet_w32' else hrnet_w48)(pretrained=True, downsample=True,
                                                                                  use_conv=(use_conv == 'conv'))
            self._backbone_id = 32 if backbone == 'hrnet_w32' else 48
        elif backbone in _resnet_family():                       # eval(backbone)(pretrained=True), hmr.py:53
            ctor, self._backbone_id = _resnet_family()[backbone]
            self.backbone = ctor(pretrained=True)
        else:
            raise NotImplementedError(f'backbone {backbone!r}: resnet18 / 34 / 50 / 101 / 152, hrnet_w32-(conv|interp), '
                                      'hrnet_w48-(conv|interp) are built (pare also ships a mobilenet trunk: not built)')
        self.use_cam_feats = use_cam_feats
        self.head = HMRHeadParams(get_backbone_info(backbone)['n_output_channels'], use_cam_feats, estimate_var=self.estimate_var,
                                  use_separate_var_branch=self.use_separate_var_branch)
        self.use_cam = use_cam
        self.smpl = SMPLHeadParams(img_res=img_res, focal_length=focal_length)
        self.img_res = img_res
        self.focal_length = focal_length
        super().train(False)
        if pretrained is not None:
            if pretrained == 'data/model_checkpoint.pt':
                self.load_pretrained_spin(pretrained)
            else:
                self.load_pretrained(pretrained)

    def _options(self):
        return {'use_cam': int(self.use_cam), 'use_cam_feats': int(self.use_cam_feats),
                'img_res': int(self.img_res), 'focal_length': float(self.focal_length),
                'backbone': self._backbone_id, 'hrnet_use_conv': self._hrnet_use_conv,
                'estimate_var': int(self.estimate_var), 'uncertainty_activation': UNCERTAINTY_ACTIVATIONS[self.uncertainty_activation]}

    def _engine_state(self, sd):
        """The library takes the variance decoders in the separate-branch layout: the doubled decoders are split here."""
        if self.estimate_var and not self.use_separate_var_branch:
            sd = dict(sd)
            for name, n in (('decpose', 144), ('decshape', 10)):
                for part in ('weight', 'bias'):
                    full = sd[f'head.{name}.{part}']
                    sd[f'head.{name}.{part}'], sd[f'head.{name}_var.{part}'] = full[:n], full[n:]
        return sd

    def _smpl_model(self):
        return self.smpl.smpl.as_model()

    @torch.no_grad()
    def forward(self, images, cam_rotmat=None, cam_intrinsics=None, bbox_scale=None, bbox_center=None,
                img_w=None, img_h=None):
        eng = self.engine(images.device)
        if self.use_cam:
            out = eng.hmr_forward(images, cam_rotmat, cam_intrinsics, bbox_scale, bbox_center, img_w, img_h)
        elif self.use_cam_feats:
            out = eng.hmr_forward(images, cam_rotmat, cam_intrinsics, None, None, None, img_h)
        else:
            out = eng.hmr_forward(images)
        if self.estimate_var:        # pare HMRHead: the two extra keys spec/losses.py:61-62 reads
            out['pred_pose_var'], out['pred_shape_var'] = eng.hmr_uncertainty(images.shape[0])
        return out

    # spec/models/hmr.py:124-136
    def load_pretrained(self, file):
        from .checkpoint import load_pretrained_model
        state_dict = torch.load(file, map_location='cpu')
        self.backbone.load_state_dict(state_dict, strict=False)
        load_pretrained_model(self.head, state_dict=state_dict, strict=False, overwrite_shape_mismatch=True)

    def load_pretrained_spin(self, file):
        state_dict = torch.load(file, map_location='cpu')['model']
        self.backbone.load_state_dict(state_dict, strict=False)
        self.head.load_state_dict(state_dict, strict=False)
