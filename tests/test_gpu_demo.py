"""The path as the reference's demo runs it (scripts/spec_demo.py): CamCalib on the FULL frame at short side 600
(camcalib/pano_dataset.py:156-162, scripts/camcalib_demo.py:95-129) once per frame, SPEC on the K crops of that frame with the
frame's camera (spec/tester.py:86-88,109-151).  Batched full-resolution CamCalib == per-frame (bit for bit), == the CPU oracle at
one 600 x N size; the one-step DemoPipeline == the per-frame composition of its parts."""
import numpy as np
import pytest
import torch

from spec_amd import synth
from tests.util import gpu_models, oracle_models, pinned_plan, rel_err, t

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def models():
    return gpu_models(True, True, DEV)


def _frames(seed, F, H, W):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (F, H // 8 + 1, W // 8 + 1, 3), dtype=np.uint8)       # blocky content: the resize has something to filter
    fr = np.repeat(np.repeat(base, 8, 1), 8, 2)[:, :H, :W]
    fr = (fr.astype(np.int32) + rng.integers(-20, 20, fr.shape)).clip(0, 255).astype(np.uint8)
    return torch.from_numpy(np.ascontiguousarray(fr))


@pytest.mark.parametrize('plan', ['throughput', 'latency'])
def test_batched_full_resolution_camcalib_equals_per_frame(models, plan):
    """F frames of 540 x 960 -> Resize(600) = 600 x 1066 (the 1080p geometry: final map 19 x 34, M no tile multiple) in ONE
    CamCalib call == one call per frame, bit for bit - the batched transform writes the same pixels and an image's logits do
    not depend on the batch within a plan (under 'auto' one 600 x 1066 frame - 12.7 crops' worth of rows - takes the latency plan and
    three take the throughput plan: equal to fp32 rounding only, checked at the end)."""
    from spec_amd.preprocess import camcalib_transform, camcalib_transform_batch
    cc, _ = models
    F = 3
    frames = _frames(5, F, 540, 960).to(DEV)
    with pinned_plan(plan, cc):
        x = camcalib_transform_batch(frames, 600)
        assert tuple(x.shape) == (F, 3, 600, 1066)
        batched = [l.clone() for l in cc(x)]
        for f in range(F):
            xf = camcalib_transform(frames[f], 600)
            assert torch.equal(xf[0], x[f])
            one = cc(xf)
            for a, b in zip(one, batched):
                assert torch.equal(a[0], b[f]), (plan, f)
    if plan == 'throughput':      # the default plan switches between 1 and 3 frames of this size: same logits to fp32 rounding
        one = cc(camcalib_transform(frames[0], 600))
        for a, b in zip(one, batched):
            assert rel_err(a[0].cpu().numpy(), b[0].cpu().numpy()) < 1e-5


def test_full_resolution_camcalib_vs_oracle(models):
    """One 600 x 800 frame (a 4:3 source at short side 600) through the CPU fp32 oracle and through both GPU plans."""
    cc, _ = models
    occ, _ = oracle_models(True, True)
    x = t(synth.images(91, 1, 600, 800))
    ref = occ(x)
    for plan in ('latency', 'throughput'):
        with pinned_plan(plan, cc):
            lg = cc(x.to(DEV))
        for a, b in zip(lg, ref):
            assert a.shape == b.shape == (1, 256)
            assert rel_err(a.cpu().numpy(), b.numpy()) < 5e-5, plan
    from spec_amd.cam_utils import convert_preds_to_angles
    ang = convert_preds_to_angles(*lg, loss_type='softargmax_l2')
    assert all(torch.isfinite(a).all() for a in ang)


@pytest.mark.parametrize('graph', [False, True])
def test_demo_pipeline_equals_per_frame_composition(models, graph):
    """DemoPipeline (one step for F frames x K detections, CamCalib beside the SPEC trunk) against the reference's structure done
    with the same parts: per frame resize -> CamCalib -> decode -> crops -> HMR with that frame's (R, K).  Plan pinned
    ('throughput'): bit-identical; also replayed as a hipGraph."""
    from spec_amd.pipeline import DemoPipeline, GraphedStep
    from spec_amd.preprocess import camcalib_transform, crop_detections
    from spec_amd import cam_utils
    cc, hm = models
    F, K, H, W = 3, 4, 360, 640
    frames = _frames(9, F, H, W).to(DEV)
    rng = np.random.default_rng(2)
    boxes = torch.from_numpy(np.stack([rng.uniform(0, W, F * K), rng.uniform(0, H, F * K), rng.uniform(60, 250, F * K),
                                       rng.uniform(120, 340, F * K)], 1).astype(np.float32)).to(DEV)
    fidx = (torch.arange(F * K) // K).to(torch.int32).to(DEV)
    keys = ('smpl_vertices', 'smpl_joints3d', 'smpl_joints2d', 'pred_cam_t', 'pred_pose', 'pred_shape')
    with pinned_plan('throughput', cc, hm):
        dp = DemoPipeline(cc, hm)
        run = GraphedStep(dp, frames, boxes, fidx) if graph else dp
        out = run(frames, boxes, fidx)
        out = {k: v.clone() for k, v in out.items()}
        if graph:                                   # a second replay on other frames, then back: no state leaks between replays
            other = run(_frames(10, F, H, W).to(DEV), boxes, fidx)['smpl_vertices'].clone()
            assert not torch.equal(other, out['smpl_vertices'])
            again = run(frames, boxes, fidx)
            assert torch.equal(again['smpl_vertices'], out['smpl_vertices'])
        assert tuple(out['cam_rotmat'].shape) == (F, 3, 3) and tuple(out['smpl_vertices'].shape) == (F * K, 6890, 3)
        for f in range(F):
            lg = cc(camcalib_transform(frames[f], 600))
            cam = cam_utils.decode_camera(lg[0], lg[1], lg[2], img_h=torch.tensor([float(H)], device=DEV), img_w=torch.tensor([float(W)], device=DEV))
            assert torch.equal(cam['cam_rotmat'][0], out['cam_rotmat'][f]) and torch.equal(cam['cam_intrinsics'][0], out['cam_intrinsics'][f])
            crops = crop_detections(frames[f], boxes[f * K:(f + 1) * K], scale=1.0, crop_size=224)
            ref = hm(crops['inp_images'], cam_rotmat=cam['cam_rotmat'].expand(K, 3, 3).contiguous(),
                     cam_intrinsics=cam['cam_intrinsics'].expand(K, 3, 3).contiguous(), bbox_scale=crops['bbox_scale'],
                     bbox_center=crops['bbox_center'], img_w=torch.full((K,), float(W), device=DEV), img_h=torch.full((K,), float(H), device=DEV))
            for k in keys:
                assert torch.equal(ref[k], out[k][f * K:(f + 1) * K]), (f, k)
    # default plan: the per-frame composition takes the latency plan (K = 4 crops, 1 frame), the step the throughput / latency
    # plan by its own batch sizes - same results to fp32 rounding
    auto = DemoPipeline(cc, hm)(frames, boxes, fidx)
    for k in keys:
        assert rel_err(auto[k].cpu().numpy(), out[k].cpu().numpy()) < 2e-5, k


def test_demo_pipeline_ragged_detections_and_empty_step(models):
    """Frames with different numbers of detections (one frame with none), detections listed out of frame order, and a step without
    any detection at all: every crop gets ITS frame's camera; an empty step returns empty outputs and the frames' cameras."""
    from spec_amd.pipeline import DemoPipeline
    cc, hm = models
    F, H, W = 3, 300, 420
    frames = _frames(21, F, H, W).to(DEV)
    rng = np.random.default_rng(4)
    fidx_host = np.array([2, 0, 2, 2, 0], dtype=np.int32)                 # frame 1 has no detection
    n = len(fidx_host)
    boxes = torch.from_numpy(np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n), rng.uniform(60, 200, n), rng.uniform(100, 280, n)], 1)
                             .astype(np.float32)).to(DEV)
    with pinned_plan('throughput', cc, hm):
        dp = DemoPipeline(cc, hm)
        out = dp(frames, boxes, torch.from_numpy(fidx_host).to(DEV))
        assert tuple(out['smpl_vertices'].shape) == (n, 6890, 3) and tuple(out['cam_rotmat'].shape) == (F, 3, 3)
        # the same detections regrouped frame by frame: identical per-crop results
        order = np.argsort(fidx_host, kind='stable')
        out2 = dp(frames, boxes[torch.from_numpy(order).to(DEV)], torch.from_numpy(fidx_host[order]).to(DEV))
        for k in ('smpl_vertices', 'smpl_joints2d', 'pred_cam_t'):
            assert torch.equal(out2[k], out[k][torch.from_numpy(order).to(DEV)]), k
        # a crop's projection uses its own frame's intrinsics: re-derive joints2d from joints3d, cam_t and (R, K) of that frame
        fi = torch.from_numpy(fidx_host).long().to(DEV)
        R, K = out['cam_rotmat'][fi].double(), out['cam_intrinsics'][fi].double()
        P = torch.einsum('bij,bkj->bki', R, out['smpl_joints3d'].double()) + out['pred_cam_t'].double()[:, None]
        P = P / P[..., 2:3]
        p2 = torch.einsum('bij,bkj->bki', K, P)[..., :2]
        assert ((p2 - out['smpl_joints2d'].double()).abs().max() / out['smpl_joints2d'].abs().max()) < 1e-5
        empty = dp(frames, boxes[:0], torch.zeros(0, dtype=torch.int32, device=DEV))
        assert tuple(empty['smpl_vertices'].shape) == (0, 6890, 3) and torch.equal(empty['cam_rotmat'], out['cam_rotmat'])
