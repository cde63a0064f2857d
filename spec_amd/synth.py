"""Deterministic synthetic parameters and inputs for the SPEC hot path.

No checkpoints, SMPL model files or datasets can be shipped (licence-gated ~1 GB download,
reference ``scripts/prepare_data.sh:4-11``), so every test and the benchmark run on
*synthetic* tensors of the real shapes.  The generator is a counter-based SplitMix64 hash
evaluated with NumPy integer arithmetic plus IEEE add/multiply only (no libm calls), so the
same seed gives bit-identical float32 tensors on any host - the golden fixtures under
``tests/golden`` store only seeds + expected outputs.

State-dict key names follow the layouts in SURVEY.md App. C (torchvision ResNet-50 trunk,
``fc_vfov/fc_pitch/fc_roll`` for CamCalib, ``head.*`` for the HMR regressor).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from . import constants as C

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """SplitMix64 finaliser on a uint64 array (wrapping arithmetic)."""
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _stream_key(seed: int, name: str) -> np.uint64:
    """Mix an integer seed with a tensor name into a 64-bit stream key (FNV-1a + SplitMix)."""
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    h ^= (seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    return _splitmix64(np.array([h], dtype=np.uint64))[0]


def uniform01(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n float64 values in [0,1) with 24 random mantissa bits (exactly representable in fp32)."""
    key = _stream_key(seed, name)
    with np.errstate(over='ignore'):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(8) + np.uint64(lane)) & _M64
        bits = _splitmix64(ctr ^ key)
    return (bits >> np.uint64(40)).astype(np.float64) * (1.0 / 16777216.0)


def uniform(seed, name, shape, lo=0.0, hi=1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed, name, shape, std=1.0, mean=0.0) -> np.ndarray:
    """Approximately normal (Irwin-Hall of 4 uniforms, unit variance) - adds/multiplies only."""
    n = int(np.prod(shape)) if len(shape) else 1
    s = uniform01(seed, name, n, 0) + uniform01(seed, name, n, 1) \
        + uniform01(seed, name, n, 2) + uniform01(seed, name, n, 3)
    z = (s - 2.0) * 1.7320508075688772
    return (mean + std * z).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# ResNet-50 trunk (torchvision v1.5 layout, no avgpool / fc)
# --------------------------------------------------------------------------------------

RESNET50_BLOCKS = (3, 4, 6, 3)
RESNET50_PLANES = (64, 128, 256, 512)


RESNET_FAMILY = {'resnet18': ('basic', (2, 2, 2, 2)), 'resnet34': ('basic', (3, 4, 6, 3)), 'resnet50': ('bottleneck', (3, 4, 6, 3)),
                 'resnet101': ('bottleneck', (3, 4, 23, 3)), 'resnet152': ('bottleneck', (3, 8, 36, 3))}


def resnet50_conv_specs(blocks=RESNET50_BLOCKS):
    """List of (name, cin, cout, k, stride, pad, bn_name) in state-dict order (Bottleneck trunks: 50 / 101 / 152 by ``blocks``)."""
    specs = [('conv1', 3, 64, 7, 2, 3, 'bn1')]
    inplanes = 64
    for li, (nb, planes) in enumerate(zip(blocks, RESNET50_PLANES), start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f'layer{li}.{b}'
            specs.append((f'{p}.conv1', inplanes, planes, 1, 1, 0, f'{p}.bn1'))
            specs.append((f'{p}.conv2', planes, planes, 3, stride, 1, f'{p}.bn2'))
            specs.append((f'{p}.conv3', planes, planes * 4, 1, 1, 0, f'{p}.bn3'))
            if b == 0:
                specs.append((f'{p}.downsample.0', inplanes, planes * 4, 1, stride, 0,
                              f'{p}.downsample.1'))
            inplanes = planes * 4
    return specs


def student_t3(seed, name, shape) -> np.ndarray:
    """Heavy-tailed unit-variance draws: Student-t with 3 degrees of freedom, z0 / sqrt((z1^2 + z2^2 + z3^2) / 3) / sqrt(3),
    from the Irwin-Hall normals above (IEEE sqrt / divide are exactly rounded, so still bit-reproducible across hosts).
    The tail is clipped at 12 standard deviations (a handful of weights per trunk)."""
    n = int(np.prod(shape)) if len(shape) else 1
    z = [normal(seed, f'{name}.t{i}', (n,)).astype(np.float64) for i in range(4)]
    chi = (z[1] * z[1] + z[2] * z[2] + z[3] * z[3]) / 3.0
    t = z[0] / np.sqrt(np.maximum(chi, 1e-3)) / 1.7320508075688772
    return np.clip(t, -12.0, 12.0).astype(np.float32).reshape(shape)


def log_uniform_pow2(seed, name, n, lo_exp=-13, hi_exp=6) -> np.ndarray:
    """n float64 values (1 + u) * 2^k, k uniform integer in [lo_exp, hi_exp]: log-uniform over [1.2e-4, 128) octave by octave,
    built with ldexp only (no libm pow / exp)."""
    k = np.floor(uniform01(seed, name + '.exp', n) * (hi_exp - lo_exp + 1)).astype(np.int64) + lo_exp
    return np.ldexp(1.0 + uniform01(seed, name + '.man', n), k)


def _pretrained_like_conv_bn(sd, seed, prefix, name, bn, cin, cout, k):
    """One convolution + BatchNorm of a 'pretrained_like' trunk before calibration: running_var log-uniform
    over six decades, gamma ~ N(0.5, 0.4) with ~5 % exact zeros, ~10 % negative and 1 % six times louder, beta ~ N(0, 0.5), Student-t(3) filters,
    ~0.5 % (at least one) all-zero filters.  ``_calibrate_trunk`` then sizes every filter and sets running_mean."""
    var = log_uniform_pow2(seed, bn + '.running_var', cout)
    fscale = np.sqrt(var / (cin * k * k))
    w = student_t3(seed, name + '.weight', (cout, cin, k, k)).astype(np.float64) * fscale.reshape(cout, 1, 1, 1)
    dead = uniform01(seed, name + '.dead', cout) < 0.005
    dead[int(uniform01(seed, name + '.dead1', 1)[0] * cout)] = True
    w[dead] = 0.0
    gamma = normal(seed, bn + '.weight', (cout,), std=0.4, mean=0.5).astype(np.float64)
    gamma[uniform01(seed, bn + '.zero', cout) < 0.05] = 0.0
    gamma[uniform01(seed, bn + '.outlier', cout) < 0.01] *= 6.0          # the few very loud channels every trained trunk has
    sd[f'{prefix}{name}.weight'] = w.astype(np.float32)
    sd[f'{prefix}{bn}.weight'] = gamma.astype(np.float32)
    sd[f'{prefix}{bn}.bias'] = normal(seed, bn + '.bias', (cout,), std=0.5)
    sd[f'{prefix}{bn}.running_mean'] = np.zeros(cout, np.float32)
    sd[f'{prefix}{bn}.running_var'] = var.astype(np.float32)
    sd[f'{prefix}{bn}.num_batches_tracked'] = np.array(0, dtype=np.int64)


def _calibrate_trunk(sd, seed, prefix, specs, basic):
    """What training does to a released checkpoint, done to the random trunk: one float64 pass over four calibration crops, layer
    by layer; each filter is scaled by the POWER OF TWO that brings the standard deviation of its pre-BN channel (over batch and
    pixels) within [0.71, 1.41] of sqrt(running_var), and running_mean is set to the channel's mean plus a N(0, 0.3 sigma) offset,
    rounded to sigma / 32.  The measured statistics only enter through those two coarse roundings, so hosts whose float64
    convolutions differ in the last bits still produce bit-identical parameters (a channel would have to sit within ~1e-12 of a
    rounding boundary), and power-of-two scaling keeps the fp32 filters exact.  Dead filters (pre-BN identically 0) keep a random
    running_mean.  torch is used for the float64 convolutions only."""
    import torch
    import torch.nn.functional as F
    x = torch.from_numpy(images(seed + 7919, 4, saturate=True)).double()

    def conv_bn(x, spec, relu):
        name, _cin, cout, _k, stride, pad, bn = spec
        w = torch.from_numpy(sd[f'{prefix}{name}.weight']).double()
        y = F.conv2d(x, w, stride=stride, padding=pad)
        yc = y.transpose(0, 1).reshape(cout, -1)
        mean, std = yc.mean(dim=1).numpy(), yc.std(dim=1, unbiased=False).numpy()
        var = sd[f'{prefix}{bn}.running_var'].astype(np.float64)
        sigma = np.sqrt(var)
        live = std > 0
        m, e = np.frexp(np.where(live, sigma / np.where(live, std, 1.0), 1.0))       # ratio = m * 2^e, m in [0.5, 1)
        q = np.ldexp(1.0, np.where(m >= 0.7071067811865476, e, e - 1))
        q = np.where(live, q, 1.0)
        xi = normal(seed, bn + '.running_mean', (cout,)).astype(np.float64)
        rm = np.where(live, np.round((mean * q / sigma + 0.3 * xi) * 32.0) / 32.0, xi) * sigma
        sd[f'{prefix}{name}.weight'] = (sd[f'{prefix}{name}.weight'].astype(np.float64) * q.reshape(-1, 1, 1, 1)).astype(np.float32)
        sd[f'{prefix}{bn}.running_mean'] = rm.astype(np.float32)
        g = torch.from_numpy(sd[f'{prefix}{bn}.weight'].astype(np.float64))
        b = torch.from_numpy(sd[f'{prefix}{bn}.bias'].astype(np.float64))
        rm32 = torch.from_numpy(sd[f'{prefix}{bn}.running_mean'].astype(np.float64))
        y = (y * torch.from_numpy(q).view(1, -1, 1, 1) - rm32.view(1, -1, 1, 1)) / torch.from_numpy(np.sqrt(var + 1e-5)).view(1, -1, 1, 1) \
            * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
        return y.clamp_(min=0) if relu else y

    nthreads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(nthreads, 16)))   # float64 convolutions crawl when oversubscribed (containers with a CPU quota)
    try:
        _calibrate_blocks(x, specs, basic, conv_bn, F)
    finally:
        torch.set_num_threads(nthreads)
    return sd


def _calibrate_blocks(x, specs, basic, conv_bn, F):
    x = F.max_pool2d(conv_bn(x, specs[0], True), 3, 2, 1)
    blocks = OrderedDict()                          # 'layer1.0' -> {member name: spec}
    for s_ in specs[1:]:
        blocks.setdefault('.'.join(s_[0].split('.')[:2]), {})[s_[0].split('.', 2)[2]] = s_
    for members in blocks.values():
        out = conv_bn(x, members['conv1'], True)
        if basic:
            out = conv_bn(out, members['conv2'], False)
        else:
            out = conv_bn(conv_bn(out, members['conv2'], True), members['conv3'], False)
        identity = conv_bn(x, members['downsample.0'], False) if 'downsample.0' in members else x
        x = (out + identity).clamp_(min=0)


def _pretrained_like_trunk(seed, prefix, specs, basic=False):
    """Bottleneck or BasicBlock trunk with the statistics of a released checkpoint (``_pretrained_like_conv_bn`` then
    ``_calibrate_trunk``); ~20 s of float64 CPU convolutions per trunk, cached per (seed, architecture) in this process."""
    key = (seed, prefix, tuple(s[0] for s in specs), basic)
    if key not in _PL_CACHE:
        sd = OrderedDict()
        for (name, cin, cout, k, _s, _p, bn) in specs:
            _pretrained_like_conv_bn(sd, seed, prefix, name, bn, cin, cout, k)
        _PL_CACHE[key] = _calibrate_trunk(sd, seed, prefix, specs, basic)
    return OrderedDict((k, v.copy()) for k, v in _PL_CACHE[key].items())


_PL_CACHE = {}
STATS = ('benign', 'pretrained_like')


def resnet50_state(seed: int, prefix: str = '', blocks=RESNET50_BLOCKS, stats: str = 'benign') -> 'OrderedDict[str, np.ndarray]':
    """Random ResNet-50 (101 / 152 by ``blocks``) trunk parameters.  ``stats='benign'``: activations kept O(1) through the blocks
    (BN variances in [0.8, 1.2]); ``'pretrained_like'``: the statistics of a released checkpoint (``_pretrained_like_conv_bn``)."""
    if stats not in STATS:
        raise ValueError(f'stats must be one of {STATS}')
    if stats == 'pretrained_like':
        return _pretrained_like_trunk(seed, prefix, resnet50_conv_specs(blocks))
    sd = OrderedDict()
    for (name, cin, cout, k, _s, _p, bn) in resnet50_conv_specs(blocks):
        fan_in = cin * k * k
        sd[f'{prefix}{name}.weight'] = normal(seed, name + '.weight', (cout, cin, k, k),
                                               std=math.sqrt(2.0 / fan_in))
        if bn.endswith('bn3'):
            g0 = 0.25          # damp the residual branch
        elif bn.endswith('downsample.1'):
            g0 = 0.7           # no ReLU follows: halve the variance gain
        else:
            g0 = 1.0
        sd[f'{prefix}{bn}.weight'] = (g0 * (1.0 + 0.1 * normal(seed, bn + '.weight', (cout,)))).astype(np.float32)
        sd[f'{prefix}{bn}.bias'] = normal(seed, bn + '.bias', (cout,), std=0.05)
        sd[f'{prefix}{bn}.running_mean'] = normal(seed, bn + '.running_mean', (cout,), std=0.1)
        sd[f'{prefix}{bn}.running_var'] = uniform(seed, bn + '.running_var', (cout,), 0.8, 1.2)
        sd[f'{prefix}{bn}.num_batches_tracked'] = np.array(0, dtype=np.int64)
    return sd


def resnet34_conv_specs(blocks=RESNET50_BLOCKS):
    """(name, cin, cout, k, stride, pad, bn_name) of the torchvision ResNet-34 (ResNet-18 by ``blocks``) trunk in state-dict order."""
    specs = [('conv1', 3, 64, 7, 2, 3, 'bn1')]
    inplanes = 64
    for li, (nb, planes) in enumerate(zip(blocks, RESNET50_PLANES), start=1):
        for b in range(nb):
            stride = 2 if (b == 0 and li > 1) else 1
            p = f'layer{li}.{b}'
            specs.append((f'{p}.conv1', inplanes, planes, 3, stride, 1, f'{p}.bn1'))
            specs.append((f'{p}.conv2', planes, planes, 3, 1, 1, f'{p}.bn2'))
            if stride != 1 or inplanes != planes:
                specs.append((f'{p}.downsample.0', inplanes, planes, 1, stride, 0, f'{p}.downsample.1'))
            inplanes = planes
    return specs


def resnet34_state(seed: int, prefix: str = '', blocks=RESNET50_BLOCKS, stats: str = 'benign') -> 'OrderedDict[str, np.ndarray]':
    """Random ResNet-34 (18 by ``blocks``) trunk parameters, activations O(1) through the blocks (bn2 damps the residual branch);
    ``stats='pretrained_like'`` as in ``resnet50_state``."""
    if stats not in STATS:
        raise ValueError(f'stats must be one of {STATS}')
    if stats == 'pretrained_like':
        return _pretrained_like_trunk(seed, prefix, resnet34_conv_specs(blocks), basic=True)
    sd = OrderedDict()
    for (name, cin, cout, k, _s, _p, bn) in resnet34_conv_specs(blocks):
        sd[f'{prefix}{name}.weight'] = normal(seed, name + '.weight', (cout, cin, k, k), std=math.sqrt(2.0 / (cin * k * k)))
        g0 = 0.25 if bn.endswith('bn2') else (0.7 if bn.endswith('downsample.1') else 1.0)
        sd[f'{prefix}{bn}.weight'] = (g0 * (1.0 + 0.1 * normal(seed, bn + '.weight', (cout,)))).astype(np.float32)
        sd[f'{prefix}{bn}.bias'] = normal(seed, bn + '.bias', (cout,), std=0.05)
        sd[f'{prefix}{bn}.running_mean'] = normal(seed, bn + '.running_mean', (cout,), std=0.1)
        sd[f'{prefix}{bn}.running_var'] = uniform(seed, bn + '.running_var', (cout,), 0.8, 1.2)
        sd[f'{prefix}{bn}.num_batches_tracked'] = np.array(0, dtype=np.int64)
    return sd


def hrnet_state(seed: int, width: int = 32, use_conv: bool = True, prefix: str = '') -> 'OrderedDict[str, np.ndarray]':
    """Random HRNet-W32 / W48 trunk parameters in the upstream key layout (``spec_amd.modules.HRNetParams``); He-normal
    convolutions, BatchNorm scales damped on the residual branches and on the exchange-unit terms so that activations
    stay O(1) through the 8 modules."""
    import torch.nn as nn
    from .modules import HRNetParams
    net = HRNetParams(width, use_conv)
    sd = OrderedDict()
    for name, m in net.named_modules():
        if isinstance(m, nn.Conv2d):
            cout, cin, k, _ = m.weight.shape
            sd[f'{prefix}{name}.weight'] = normal(seed, name + '.weight', (cout, cin, k, k), std=math.sqrt(2.0 / (cin * k * k)))
        elif isinstance(m, nn.BatchNorm2d):
            n = m.num_features
            last = name.rsplit('.', 1)[-1]
            if 'fuse_layers' in name:
                # chain members followed by a ReLU keep gain 1; the member that enters the sum is damped
                parts = name.split('.')
                i, j = int(parts[3]), int(parts[4])
                if j < i:
                    k = int(parts[5])
                    g0 = 0.3 if k == i - j - 1 else 1.0
                else:
                    g0 = 0.3
            elif last in ('bn2', 'bn3') and ('branches' in name or name.startswith('layer1')):
                g0 = 0.25 if (last == 'bn3' or 'branches' in name) else 1.0
            elif name.endswith('downsample.1'):
                g0 = 0.7
            else:
                g0 = 1.0
            sd[f'{prefix}{name}.weight'] = (g0 * (1.0 + 0.1 * normal(seed, name + '.weight', (n,)))).astype(np.float32)
            sd[f'{prefix}{name}.bias'] = normal(seed, name + '.bias', (n,), std=0.05)
            sd[f'{prefix}{name}.running_mean'] = normal(seed, name + '.running_mean', (n,), std=0.1)
            sd[f'{prefix}{name}.running_var'] = uniform(seed, name + '.running_var', (n,), 0.8, 1.2)
            sd[f'{prefix}{name}.num_batches_tracked'] = np.array(0, dtype=np.int64)
    return sd


def resnet_family_state(seed: int, backbone: str, prefix: str = '', stats: str = 'benign'):
    """(trunk state, feature width) of a torchvision-family trunk by name (resnet18 / 34 / 50 / 101 / 152)."""
    kind, blocks = RESNET_FAMILY[backbone]
    if kind == 'basic':
        return resnet34_state(seed, prefix, blocks, stats), 512
    return resnet50_state(seed, prefix, blocks, stats), 2048


def camcalib_state(seed: int = 1001, fc_std: float = 0.05, nbins: int = C.NUM_CAMCALIB_BINS, backbone: str = 'resnet50',
                   num_fc_layers: int = 1, num_fc_channels: int = 1024, stats: str = 'benign'):
    """CameraRegressorNetwork parameters (camcalib/model.py:40-70 layout): one Linear per angle, or the
    ``fc_*.{0..L-1}`` Linear chain of ``_get_fc_layers``."""
    sd, feat = resnet_family_state(seed, backbone, 'backbone.', stats)
    for head in ('fc_vfov', 'fc_pitch', 'fc_roll'):
        if num_fc_layers == 1:
            sd[f'{head}.weight'] = normal(seed, head + '.weight', (nbins, feat), std=fc_std)
            sd[f'{head}.bias'] = normal(seed, head + '.bias', (nbins,), std=0.1)
            continue
        for l in range(num_fc_layers):
            nin = feat if l == 0 else num_fc_channels
            nout = nbins if l == num_fc_layers - 1 else num_fc_channels
            sd[f'{head}.{l}.weight'] = normal(seed, f'{head}.{l}.weight', (nout, nin), std=fc_std if l == num_fc_layers - 1 else math.sqrt(1.0 / nin))
            sd[f'{head}.{l}.bias'] = normal(seed, f'{head}.{l}.bias', (nout,), std=0.1)
    return sd


def _random_rot6d(seed, name, n):
    """rot6d (first two columns, row-major 3x2) of n random rotations: returns (n*6,) fp32."""
    a = normal(seed, name + '.a', (n, 3)).astype(np.float64)
    b = normal(seed, name + '.b', (n, 3)).astype(np.float64)
    a = 0.35 * a + np.array([1.0, 0.0, 0.0])
    b = 0.35 * b + np.array([0.0, 1.0, 0.0])
    # 6d layout is x.view(-1,3,2): element [i, c] = column c, row i
    out = np.stack([a, b], axis=-1)  # (n,3,2)
    return out.reshape(-1).astype(np.float32)


def _orthonormal_rot6d(seed, name, n):
    """rot6d of n uniformly random ROTATIONS (Gram-Schmidt of two normal vectors; sqrt / divide only): what a trained regressor's
    mean pose looks like to ``rot6d_to_rotmat`` - unit, orthogonal columns, far from the identity."""
    a = normal(seed, name + '.a', (n, 3)).astype(np.float64)
    b = normal(seed, name + '.b', (n, 3)).astype(np.float64)
    b1 = a / np.sqrt((a * a).sum(-1, keepdims=True))
    u2 = b - (b1 * b).sum(-1, keepdims=True) * b1
    b2 = u2 / np.sqrt((u2 * u2).sum(-1, keepdims=True))
    return np.stack([b1, b2], axis=-1).reshape(-1).astype(np.float32)


def hmr_state(seed: int = 1002, use_cam_feats: bool = True, dec_gain: float = 1.0, backbone: str = 'resnet50',
              stats: str = 'benign', cam_gain: float = None, estimate_var: bool = False, use_separate_var_branch: bool = False,
              with_trunk: bool = True):
    """HMR parameters: trunk + HMRHead (fc1, fc2, decpose, decshape, deccam, init_*).  ``backbone``: 'resnet50' or
    'hrnet_w32-conv' / 'hrnet_w32-interp' / 'hrnet_w48-...' (spec/models/hmr.py:44-53).  ``dec_gain`` 4 is Xavier gain 1 on the
    decoders; ``cam_gain`` (default: ``dec_gain``) sizes ``deccam`` alone - a trained regressor keeps the weak-perspective scale
    away from 0, where tz = 2f / (res * s) is ill-conditioned for the reference's own fp32 arithmetic too."""
    if backbone.startswith('hrnet'):
        name, mode = backbone.split('-')
        width = 32 if name == 'hrnet_w32' else 48
        sd = hrnet_state(seed, width, mode == 'conv', 'backbone.') if with_trunk else OrderedDict()
        feat = width * 15
    elif not with_trunk:            # head parameters only (the caller brings its own trunk)
        sd, feat = OrderedDict(), (512 if RESNET_FAMILY[backbone][0] == 'basic' else 2048)
    else:
        sd, feat = resnet_family_state(seed, backbone, 'backbone.', stats)
    nin = feat + 144 + 13 + (7 if use_cam_feats else 0)

    def linear(name, nout, nin_, bound=None, bias_bound=None):
        bound = (1.0 / math.sqrt(nin_)) if bound is None else bound
        bias_bound = (1.0 / math.sqrt(nin_)) if bias_bound is None else bias_bound
        sd[f'head.{name}.weight'] = uniform(seed, f'head.{name}.weight', (nout, nin_), -bound, bound)
        sd[f'head.{name}.bias'] = uniform(seed, f'head.{name}.bias', (nout,), -bias_bound, bias_bound)

    linear('fc1', 1024, nin)
    linear('fc2', 1024, 1024)
    # upstream uses xavier_uniform(gain=0.01); a larger gain makes the synthetic poses vary
    # between images so that parity tests exercise the full rot6d / LBS range.
    for name, nout in (('decpose', 144), ('decshape', 10), ('deccam', 3)):
        gain = cam_gain if (name == 'deccam' and cam_gain is not None) else dec_gain
        xav = gain * 0.25 * math.sqrt(6.0 / (1024 + nout))
        linear(name, nout, 1024, bound=xav, bias_bound=0.01)
    if estimate_var:
        # pare HMRHead with estimate_var (spec/models/hmr.py:35-38,59-61): variance decoders as their own layers, or as the second
        # half of doubled decpose / decshape (the mean half keeps the values above)
        for name, nout in (('decpose', 144), ('decshape', 10)):
            xav = 4.0 * 0.25 * math.sqrt(6.0 / (1024 + nout))
            linear(name + '_var', nout, 1024, bound=xav, bias_bound=0.3)
            if not use_separate_var_branch:
                for part in ('weight', 'bias'):
                    sd[f'head.{name}.{part}'] = np.concatenate([sd[f'head.{name}.{part}'], sd.pop(f'head.{name}_var.{part}')], axis=0)
    sd['head.init_pose'] = (_orthonormal_rot6d if stats == 'pretrained_like' else _random_rot6d)(seed, 'head.init_pose', 24).reshape(1, 144)
    sd['head.init_shape'] = normal(seed, 'head.init_shape', (1, 10), std=0.5)
    sd['head.init_cam'] = np.array([[0.9, 0.0, 0.0]], dtype=np.float32) \
        + normal(seed, 'head.init_cam', (1, 3), std=0.02)
    return sd


# --------------------------------------------------------------------------------------
# SMPL-shaped body model
# --------------------------------------------------------------------------------------

def smpl_model(seed: int = 1003, nv: int = C.NUM_SMPL_VERTS):
    """A synthetic body model with the tensor shapes/dtypes of smplx.SMPL (+ SPIN extras).

    Keys: v_template (nv,3), shapedirs (nv,3,10), posedirs (207, nv*3), J_regressor (24,nv),
    lbs_weights (nv,24), J_regressor_extra (9,nv), parents (24,) int32,
    extra_vertex_ids (21,) int32, joint_map (49,) int32.
    """
    m = OrderedDict()
    box = np.array([0.6, 1.7, 0.3], dtype=np.float64)
    vt = (uniform01(seed, 'v_template', nv * 3).reshape(nv, 3) - 0.5) * box
    m['v_template'] = vt.astype(np.float32)
    m['shapedirs'] = normal(seed, 'shapedirs', (nv, 3, 10), std=0.01)
    m['posedirs'] = normal(seed, 'posedirs', (C.NUM_POSE_BASIS, nv * 3), std=0.002)

    def sparse_rows(name, rows, nnz):
        w = np.zeros((rows, nv), dtype=np.float64)
        idx = (uniform01(seed, name + '.idx', rows * nnz) * nv).astype(np.int64).reshape(rows, nnz)
        val = uniform01(seed, name + '.val', rows * nnz).reshape(rows, nnz) + 0.05
        for r in range(rows):
            np.add.at(w[r], idx[r], val[r])
        w /= w.sum(axis=1, keepdims=True)
        return w.astype(np.float32)

    m['J_regressor'] = sparse_rows('J_regressor', 24, 48)
    m['J_regressor_extra'] = sparse_rows('J_regressor_extra', C.NUM_EXTRA_REGRESSED, 32)

    lw = np.zeros((nv, 24), dtype=np.float64)
    jidx = (uniform01(seed, 'lbs.idx', nv * 4) * 24).astype(np.int64).reshape(nv, 4)
    jval = uniform01(seed, 'lbs.val', nv * 4).reshape(nv, 4) + 0.05
    np.add.at(lw, (np.repeat(np.arange(nv), 4), jidx.reshape(-1)), jval.reshape(-1))
    lw /= lw.sum(axis=1, keepdims=True)
    m['lbs_weights'] = lw.astype(np.float32)

    m['parents'] = np.array(C.SMPL_PARENTS, dtype=np.int32)
    ids = np.array(C.SMPL_EXTRA_VERTEX_IDS, dtype=np.int64)
    if nv != C.NUM_SMPL_VERTS:
        ids = ids % nv
    m['extra_vertex_ids'] = ids.astype(np.int32)
    m['joint_map'] = np.array(C.JOINT_MAP49, dtype=np.int32)
    return m


def h36m_regressor(seed: int = 1003, nv: int = C.NUM_SMPL_VERTS) -> np.ndarray:
    """A synthetic 17-row joint regressor standing in for ``data/J_regressor_h36m.npy``
    (used by the evaluation metrics, spec/trainer.py:96-99,272-279): rows sum to 1."""
    J = smpl_model(seed, nv)['J_regressor']
    return np.concatenate([J[:12], J[12:17] * 0.5 + J[17:22] * 0.5], 0).astype(np.float32)


# --------------------------------------------------------------------------------------
# inputs
# --------------------------------------------------------------------------------------

def images(seed: int, batch: int, h: int = 224, w: int = 224, saturate: bool = False) -> np.ndarray:
    """(B,3,H,W) fp32 crops: uniform[0,1) pixels, ImageNet-normalised (spec/constants.py:20-21); ``saturate`` adds clipped
    black / white rectangles."""
    x = uniform01(seed, 'images', batch * 3 * h * w).reshape(batch, 3, h, w)
    # per-image contrast / per-channel brightness / a horizontal ramp, so that images (and the
    # features, camera angles and poses regressed from them) differ from one another
    gain = 0.25 + 0.75 * uniform01(seed, 'images.gain', batch).reshape(batch, 1, 1, 1)
    offs = 0.5 * uniform01(seed, 'images.offset', batch * 3).reshape(batch, 3, 1, 1)
    ramp = (uniform01(seed, 'images.ramp', batch).reshape(batch, 1, 1, 1) - 0.5) \
        * np.linspace(-1.0, 1.0, w).reshape(1, 1, 1, w)
    x = np.clip(x * gain + offs * (1.0 - gain) + 0.5 * ramp, 0.0, 1.0)
    if saturate:
        # two rectangles per image clipped to pure black / pure white (blown highlights, letter-box bars of a real frame):
        # constant regions whose stem and Winograd input tiles carry no variation at all
        u = uniform01(seed, 'images.sat', batch * 8).reshape(batch, 2, 4)
        for b in range(batch):
            for j in range(2):
                y0, x0 = int(u[b, j, 0] * h * 0.7), int(u[b, j, 1] * w * 0.7)
                y1, x1 = y0 + 8 + int(u[b, j, 2] * h * 0.3), x0 + 8 + int(u[b, j, 3] * w * 0.3)
                x[b, :, y0:y1, x0:x1] = float(j)
    mean = np.array(C.IMG_NORM_MEAN).reshape(1, 3, 1, 1)
    std = np.array(C.IMG_NORM_STD).reshape(1, 3, 1, 1)
    return ((x - mean) / std).astype(np.float32)


def bbox_inputs(seed: int, batch: int, img_w: float = 224.0, img_h: float = 224.0, jitter: bool = True):
    """Per-image scalars in the tester's convention (spec/tester.py:127-134):
    bbox_scale = bbox_h / 200, bbox_center = (cx, cy), img_w, img_h."""
    if jitter:
        scale = uniform(seed, 'bbox_scale', (batch,), 0.8, 1.4)
        center = np.stack([uniform(seed, 'bbox_cx', (batch,), 0.35 * img_w, 0.65 * img_w),
                           uniform(seed, 'bbox_cy', (batch,), 0.35 * img_h, 0.65 * img_h)], axis=1)
    else:
        scale = np.full((batch,), 224.0 / 200.0, dtype=np.float32)
        center = np.tile(np.array([[img_w / 2, img_h / 2]], dtype=np.float32), (batch, 1))
    return (scale.astype(np.float32), center.astype(np.float32),
            np.full((batch,), img_w, dtype=np.float32), np.full((batch,), img_h, dtype=np.float32))
