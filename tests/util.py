"""Shared builders for the tests (oracle side and GPU side use the same seeded tensors)."""
import contextlib
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEED_CAMCALIB, SEED_HMR, SEED_SMPL, SEED_IMG = 1001, 1002, 1003, 20210001


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    """max |a-b| / max |b|  (the '1e-4 relative fp32' criterion of BASELINE.json, tensor-wise)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


_cache = {}


def synth_states(use_cam_feats=True):
    from spec_amd import synth
    key = ('states', use_cam_feats)
    if key not in _cache:
        _cache[key] = (synth.camcalib_state(SEED_CAMCALIB), synth.hmr_state(SEED_HMR, use_cam_feats))
    return _cache[key]


def smpl_model():
    from spec_amd import synth
    if 'smpl' not in _cache:
        _cache['smpl'] = synth.smpl_model(SEED_SMPL)
    return _cache['smpl']


def oracle_models(use_cam=True, use_cam_feats=True):
    from oracle import heads
    from oracle.models import CamCalibOracle, HMROracle, load_numpy_state
    torch.set_grad_enabled(False)
    heads.set_assets(smpl_model=smpl_model())
    cs, hs = synth_states(use_cam_feats)
    cc = load_numpy_state(CamCalibOracle().eval(), cs)
    hm = load_numpy_state(HMROracle(use_cam=use_cam, use_cam_feats=use_cam_feats).eval(), hs)
    return cc, hm


def gpu_models(use_cam=True, use_cam_feats=True, device='cuda:0'):
    from spec_amd import assets
    from spec_amd.modules import HMR, CameraRegressorNetwork
    assets.use_synthetic_assets(SEED_SMPL)
    cs, hs = synth_states(use_cam_feats)
    cc = CameraRegressorNetwork()
    cc.load_state_dict({k: t(v) for k, v in cs.items()}, strict=True)
    hm = HMR(use_cam=use_cam, use_cam_feats=use_cam_feats)
    missing, unexpected = hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    assert not unexpected and all(m.startswith('smpl.') for m in missing), (missing, unexpected)
    return cc.to(device).eval(), hm.to(device).eval()


@contextlib.contextmanager
def cpu_threads(n=16):
    """Cap torch's intra-op threads for the float64 oracle passes: on the GPU boxes the container sees 128 logical CPUs but may use
    16 (cgroup quota) - float64 convolutions run 3-5 x slower oversubscribed (bench.py's thread sweep peaks at 16 for the same reason)."""
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(old, n)))
    try:
        yield
    finally:
        torch.set_num_threads(old)


@contextlib.contextmanager
def pinned_plan(plan, *modules):
    """Run a block with the trunk execution plan of ``modules`` pinned ('throughput' | 'latency' | 'auto').  Bit-identity
    across batch sizes holds WITHIN a plan; 'auto' (the default) switches to the latency plan at 10 images or fewer (16 for a single trunk)."""
    old = [m.plan for m in modules]
    for m in modules:
        m.set_plan(plan)
    try:
        yield
    finally:
        for m, o in zip(modules, old):
            m.set_plan(o)


# ---- released-checkpoint-like statistics (spec_amd.synth, stats='pretrained_like') -------------------
PL_SEED_CC, PL_SEED_HM, PL_SEED_IMG = 2101, 2102, 2103
PL_DEC_GAIN, PL_CAM_GAIN = 2.0, 1.0       # Xavier gain 0.5 on decpose / decshape around an orthonormal, far-from-identity init_pose (pose6d moves
                                          # by ~0.7 rms per element between images; rot6d_to_rotmat stays conditioned below ~8); deccam at 0.25


def pretrained_like_states():
    """(CamCalib state, HMR state): BN variances over six decades, zero / negative / loud gammas, dead filters, Student-t
    weights, calibrated running statistics, O(1)-gain decoders (~45 s of float64 CPU convolutions, once per process)."""
    from spec_amd import synth
    if 'pl_states' not in _cache:
        _cache['pl_states'] = (synth.camcalib_state(PL_SEED_CC, stats='pretrained_like'),
                               synth.hmr_state(PL_SEED_HM, True, dec_gain=PL_DEC_GAIN, cam_gain=PL_CAM_GAIN, stats='pretrained_like'))
    return _cache['pl_states']


def pl_oracle_models(double=False):
    from oracle import heads
    from oracle.models import CamCalibOracle, HMROracle, load_numpy_state
    torch.set_grad_enabled(False)
    heads.set_assets(smpl_model=smpl_model())
    cs, hs = pretrained_like_states()
    cc = load_numpy_state(CamCalibOracle().eval(), cs)
    hm = load_numpy_state(HMROracle(use_cam=True, use_cam_feats=True).eval(), hs)
    return (cc.double(), hm.double()) if double else (cc, hm)


def pl_gpu_models(device='cuda:0'):
    from spec_amd import assets
    from spec_amd.modules import HMR, CameraRegressorNetwork
    assets.use_synthetic_assets(SEED_SMPL)
    cs, hs = pretrained_like_states()
    cc = CameraRegressorNetwork()
    cc.load_state_dict({k: t(v) for k, v in cs.items()}, strict=True)
    hm = HMR(use_cam=True, use_cam_feats=True)
    missing, unexpected = hm.load_state_dict({k: t(v) for k, v in hs.items()}, strict=False)
    assert not unexpected and all(m.startswith('smpl.') for m in missing), (missing, unexpected)
    return cc.to(device).eval(), hm.to(device).eval()


def per_channel_errors(x, ref64):
    """Per-channel max-norm error of an (..., C) map against a float64 reference of the same shape -> (C,) float64."""
    d = np.abs(np.asarray(x, dtype=np.float64) - np.asarray(ref64, dtype=np.float64))
    return d.reshape(-1, d.shape[-1]).max(axis=0)


def float64_mesh(hmr_oracle64, images, cam_rotmat, cam_intrinsics, img_h):
    """(vertices, joints3d, head outputs) of an HMR oracle converted with ``.double()``: trunk, regressor head and SMPL in float64
    (the camera conversion / projection leaves of oracle/geometry.py cast to fp32 as upstream does, so the arbiter stops at the mesh)."""
    with cpu_threads():
        f64 = hmr_oracle64.backbone(images.double())
    vfov = 2 * torch.atan(img_h.double() / (2 * cam_intrinsics[:, 0, 0].double()))
    h64 = hmr_oracle64.head(f64, cam_rotmat=cam_rotmat.double(), cam_vfov=vfov)
    v64, j64 = hmr_oracle64.smpl.smpl(h64['pred_shape'], h64['pred_pose'])
    return v64, j64, h64


def pretrained_like_by_hooks(trunk, seed, images):
    """Released-checkpoint-like statistics for ANY oracle trunk made of Conv2d -> BatchNorm2d pairs (HRNet: only the oracle knows
    the wiring), the hook-driven twin of ``spec_amd.synth._calibrate_trunk``: Student-t(3) filters with a few dead ones, running_var
    log-uniform over six decades, gamma ~ N(0.5, 0.4) with exact zeros / negatives / loud channels, beta ~ N(0, 0.5); then ONE float64
    pass over ``images`` in which every BatchNorm, just before it runs, scales the filters of the convolution that fed it by the power
    of two that brings the channel's standard deviation within [0.71, 1.41] sigma and sets running_mean to the channel mean + N(0, 0.3
    sigma) rounded to sigma / 32.  Returns the state as {name: float32 ndarray}; ``trunk`` is left in float32 with that state."""
    from spec_amd import synth
    torch.set_grad_enabled(False)
    g = torch.Generator().manual_seed(seed)       # (the filters come from torch's generator: this state is used inside ONE process,
    for name, m in trunk.named_modules():         #  by the oracle and the GPU module alike - no cross-host reproducibility needed)
        if isinstance(m, torch.nn.Conv2d):
            cout, cin, k, _ = m.weight.shape
            z = torch.randn(4, cout, cin, k, k, generator=g)
            w = (z[0] / torch.sqrt((z[1:] ** 2).mean(0).clamp_min(1e-3)) / 3 ** 0.5).clamp_(-12, 12) * (1.0 / (cin * k * k)) ** 0.5
            w[torch.rand(cout, generator=g) < 0.005] = 0.0
            m.weight.copy_(w)
        elif isinstance(m, torch.nn.BatchNorm2d):
            n = m.num_features
            gamma = synth.normal(seed, name + '.weight', (n,), std=0.4, mean=0.5)
            gamma[synth.uniform01(seed, name + '.zero', n) < 0.05] = 0.0
            gamma[synth.uniform01(seed, name + '.outlier', n) < 0.01] *= 6.0
            m.weight.copy_(t(gamma))
            m.bias.copy_(t(synth.normal(seed, name + '.bias', (n,), std=0.5)))
            m.running_var.copy_(t(synth.log_uniform_pow2(seed, name + '.running_var', n).astype(np.float32)))
            m.running_mean.zero_()
    trunk.double()
    last, names, hooks = {}, {m: n for n, m in trunk.named_modules()}, []

    def conv_hook(mod, inp, out):
        last['conv'] = mod

    def bn_pre(mod, inp):
        y, conv = inp[0], last['conv']
        assert conv.out_channels == mod.num_features, (names[mod], names[conv])
        yc = y.transpose(0, 1).reshape(mod.num_features, -1)
        mean, std = yc.mean(dim=1).numpy(), yc.std(dim=1, unbiased=False).numpy()
        sigma = np.sqrt(mod.running_var.numpy())
        live = std > 0
        mm, e = np.frexp(np.where(live, sigma / np.where(live, std, 1.0), 1.0))
        q = np.where(live, np.ldexp(1.0, np.where(mm >= 0.7071067811865476, e, e - 1)), 1.0)
        xi = synth.normal(seed, names[mod] + '.running_mean', (mod.num_features,)).astype(np.float64)
        rm = np.where(live, np.round((mean * q / sigma + 0.3 * xi) * 32.0) / 32.0, xi) * sigma
        conv.weight.mul_(t(q).view(-1, 1, 1, 1))
        mod.running_mean.copy_(t(rm.astype(np.float32).astype(np.float64)))
        return (y * t(q).view(1, -1, 1, 1),)

    for m in trunk.modules():
        if isinstance(m, torch.nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, torch.nn.BatchNorm2d):
            hooks.append(m.register_forward_pre_hook(bn_pre))
    with cpu_threads():
        trunk(images.double())
    for h_ in hooks:
        h_.remove()
    trunk.float()
    return {k: v.detach().numpy().copy() for k, v in trunk.state_dict().items() if not k.endswith('num_batches_tracked')}
