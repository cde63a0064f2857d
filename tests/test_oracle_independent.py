"""The oracle's un-vendored leaf arithmetic against INDEPENDENT published implementations that happen to be installed
(no weights or network needed): the ResNet-50 / ResNet-34 block wiring against Hugging Face ``transformers``' port of the
torchvision v1.5 network (stride on the 3x3, downsample shortcut, BN eps, ReLU placement, max-pool), ``batch_rodrigues``
and ``batch_euler2matrix`` against ``scipy.spatial.transform.Rotation``.  This narrows "parity unpinned" for those leaves:
the restatement agrees with a second upstream, not only with itself."""
import numpy as np
import pytest
import torch

from spec_amd import synth

torch.set_grad_enabled(False)


def _hf_name(k):
    """torchvision trunk key -> transformers ResNetModel key."""
    if k.startswith('conv1.'):
        return 'embedder.embedder.convolution.' + k[len('conv1.'):]
    if k.startswith('bn1.'):
        return 'embedder.embedder.normalization.' + k[len('bn1.'):]
    p = k.split('.')                                  # layerL.b.(convK|bnK|downsample.I).param
    stage, blk = int(p[0][5:]) - 1, p[1]
    base = f'encoder.stages.{stage}.layers.{blk}.'
    if p[2] == 'downsample':
        return base + 'shortcut.' + ('convolution.' if p[3] == '0' else 'normalization.') + p[4]
    idx = int(p[2][-1]) - 1
    return base + f'layer.{idx}.' + ('convolution.' if p[2].startswith('conv') else 'normalization.') + p[3]


@pytest.mark.parametrize('depth', [50, 34])
def test_resnet_trunk_vs_hf_transformers(depth):
    tr = pytest.importorskip('transformers')
    from oracle.models import load_numpy_state
    from oracle.resnet import ResNet34Trunk, ResNet50Trunk
    if depth == 50:
        cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                              layer_type='bottleneck', hidden_act='relu', downsample_in_first_stage=False,
                              downsample_in_bottleneck=False)
        sd, trunk = synth.resnet50_state(1001), ResNet50Trunk().eval()
    else:
        cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[3, 4, 6, 3],
                              layer_type='basic', hidden_act='relu', downsample_in_first_stage=False)
        sd, trunk = synth.resnet34_state(1001), ResNet34Trunk().eval()
    hf = tr.ResNetModel(cfg).eval()
    mapped = {_hf_name(k): torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    assert set(mapped) == set(hf.state_dict().keys())              # 318 / 216 tensors, one to one
    hf.load_state_dict(mapped, strict=True)
    load_numpy_state(trunk, sd)
    x = torch.from_numpy(synth.images(21, 2))[:, :, :160, :128].contiguous()
    ref = hf(x).last_hidden_state
    out = trunk(x)
    assert out.shape == ref.shape == (2, 2048 if depth == 50 else 512, 5, 4)
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < 1e-6, err                                          # same torch ops in the same order: ~0


def test_batch_rodrigues_vs_scipy():
    from scipy.spatial.transform import Rotation
    from oracle.smpl import batch_rodrigues
    rng = np.random.default_rng(0)
    rv = rng.standard_normal((200, 3)) * rng.uniform(0.01, 3.0, (200, 1))
    R = batch_rodrigues(torch.from_numpy(rv)).numpy()              # float64 in, float64 out
    ref = Rotation.from_rotvec(rv).as_matrix()
    assert np.abs(R - ref).max() < 1e-7                             # the +1e-8 guard moves the axis by ~1e-8
    R32 = batch_rodrigues(torch.from_numpy(rv.astype(np.float32))).numpy()
    assert np.abs(R32 - ref).max() < 5e-6


def test_batch_euler2matrix_vs_scipy():
    from scipy.spatial.transform import Rotation
    from oracle.geometry import batch_euler2matrix
    rng = np.random.default_rng(1)
    ang = rng.uniform(-1.2, 1.2, (100, 3)).astype(np.float32)
    R = batch_euler2matrix(torch.from_numpy(ang)).numpy()
    ref = Rotation.from_euler('XYZ', ang.astype(np.float64)).as_matrix()     # intrinsic X-Y-Z = Rx(x) Ry(y) Rz(z)
    assert np.abs(R - ref).max() < 2e-6
    # the hand-off uses (pitch, 0, roll): Rx(pitch) Rz(roll)  (spec/utils/cam_params.py:37)
    pr = np.stack([ang[:, 0], np.zeros(100, np.float32), ang[:, 2]], 1)
    R2 = batch_euler2matrix(torch.from_numpy(pr)).numpy()
    ref2 = Rotation.from_euler('x', pr[:, 0].astype(np.float64)).as_matrix() @ Rotation.from_euler('z', pr[:, 2].astype(np.float64)).as_matrix()
    assert np.abs(R2 - ref2).max() < 2e-6


def test_procrustes_vs_scipy():
    """reconstruction_error's similarity alignment against scipy.linalg.orthogonal_procrustes (no reflection cases)."""
    from scipy.linalg import orthogonal_procrustes
    from oracle.metrics import compute_similarity_transform
    rng = np.random.default_rng(2)
    for _ in range(10):
        S2 = rng.standard_normal((14, 3))
        Rm = np.linalg.qr(rng.standard_normal((3, 3)))[0]
        if np.linalg.det(Rm) < 0:
            Rm[:, 0] *= -1
        S1 = (S2 @ Rm.T) * 1.3 + rng.standard_normal(3) + 0.01 * rng.standard_normal((14, 3))
        hat = compute_similarity_transform(S1, S2)
        A, Bm = S1 - S1.mean(0), S2 - S2.mean(0)
        Rp, sc = orthogonal_procrustes(A, Bm)
        ref = (A @ Rp) * (sc / (A ** 2).sum()) + S2.mean(0)
        assert np.abs(hat - ref).max() < 1e-9
