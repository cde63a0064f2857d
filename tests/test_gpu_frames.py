"""Real-input front end (spec/tester.py:109-151): frames in host memory -> batched crops -> the step.
The batched path must equal the per-frame path of the reference's structure BIT FOR BIT."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import gpu_models, t  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _frames_and_boxes(F, H, W, per_frame, seed):
    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    boxes, fidx = [], []
    for f in range(F):
        for _ in range(per_frame[f]):
            bw, bh = rng.uniform(60, 0.9 * W), rng.uniform(80, 1.1 * H)          # some boxes leave the frame
            boxes.append([rng.uniform(0, W), rng.uniform(0, H), bw, bh])
            fidx.append(f)
    return frames, np.asarray(boxes, np.float32).reshape(-1, 4), np.asarray(fidx, np.int32)


@pytest.mark.parametrize('F,H,W', [(3, 240, 320), (5, 135, 97), (2, 1080, 1920)])
def test_batched_crops_equal_per_frame_crops_bit_for_bit(F, H, W):
    from spec_amd.preprocess import crop_detections, crop_detections_batch
    per = [(f * 2 + 1) % 4 for f in range(F)]                       # includes frames without detections
    frames, boxes, fidx = _frames_and_boxes(F, H, W, per, 11 + F)
    slab = t(frames).to(DEV)
    got = crop_detections_batch(slab, fidx, boxes, scale=1.0, crop_size=224)
    k = 0
    for f in range(F):
        n = per[f]
        if n == 0:
            continue
        ref = crop_detections(slab[f], boxes[k:k + n], scale=1.0, crop_size=224)
        for key in ('inp_images', 'bbox_scale', 'bbox_center'):
            assert torch.equal(got[key][k:k + n], ref[key]), (f, key)
        k += n
    assert k == len(boxes)


def test_frame_index_out_of_range_is_rejected_on_the_host_and_clamped_on_the_device():
    """A stale / negative / too large frame index must never become an out-of-bounds read: a host-side index is range-checked,
    a device-side one (the host cannot look at it without a sync) is clamped into the slab by the kernel."""
    from spec_amd.preprocess import crop_detections_batch
    rng = np.random.default_rng(3)
    frames = torch.from_numpy(rng.integers(0, 256, (2, 64, 80, 3), dtype=np.uint8)).to(DEV)
    dets = torch.tensor([[40., 32., 30., 50.], [20., 20., 25., 40.], [60., 40., 20., 30.]])
    with pytest.raises(ValueError):
        crop_detections_batch(frames, torch.tensor([0, 2, 1], dtype=torch.int32), dets)
    with pytest.raises(ValueError):
        crop_detections_batch(frames, [0, -1, 1], dets)
    bad = crop_detections_batch(frames, torch.tensor([0, 7, -5], dtype=torch.int32, device=DEV), dets)
    ref = crop_detections_batch(frames, torch.tensor([0, 1, 0], dtype=torch.int32, device=DEV), dets)
    torch.cuda.synchronize()
    assert torch.equal(bad['inp_images'], ref['inp_images'])        # 7 -> last frame, -5 -> frame 0


def test_crop_into_batch_buffer_slices():
    """crop_detections(out=...) writes into slices of a larger batch buffer (what the batched tester does)."""
    from spec_amd.preprocess import crop_detections
    frames, boxes, _ = _frames_and_boxes(1, 200, 300, [5], 3)
    fr = t(frames[0]).to(DEV)
    ref = crop_detections(fr, boxes)
    buf = {'inp_images': torch.zeros(9, 3, 224, 224, device=DEV), 'bbox_scale': torch.zeros(9, device=DEV),
           'bbox_center': torch.zeros(9, 2, device=DEV)}
    crop_detections(fr, boxes, out={k_: v[2:7] for k_, v in buf.items()})
    for key in buf:
        assert torch.equal(buf[key][2:7], ref[key]) and float(buf[key][:2].abs().sum()) == 0 and float(buf[key][7:].abs().sum()) == 0
    with pytest.raises(ValueError):
        crop_detections(fr, boxes, out={k_: v[:4] for k_, v in buf.items()})


@pytest.mark.parametrize('graph', [False, True])
def test_frame_stream_equals_per_frame_forward(graph):
    """FrameStream (pinned host slab -> copy stream -> batched crops -> step, two slabs alternating) against the reference's
    structure: per frame, crops of that frame's detections, one forward per frame.  Bit-identical outputs (one execution plan:
    the step of 12 crops and the per-frame forwards of 3 would otherwise take the throughput and the latency plan)."""
    from spec_amd.frames import FrameStream
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline
    from spec_amd.preprocess import crop_detections
    cc, hm = gpu_models(True, True, DEV)
    cc.set_plan('throughput'); hm.set_plan('throughput')
    pipe = SpecPipeline(cc, hm, overlap=True)
    F, H, W, N = 4, 360, 480, 12
    per = [3, 3, 3, 3]
    dev = torch.device(DEV)
    step = pipe
    if graph:
        z = torch.zeros
        step = GraphedPipeline(pipe, z(N, 3, 224, 224, device=dev), z(N, device=dev) + 1, z(N, 2, device=dev) + 100,
                               z(N, device=dev) + W, z(N, device=dev) + H)
    fs = FrameStream(step, dev, (H, W), F, N)
    hosts = [fs.host_buffers() for _ in range(3)]
    outs = []
    for s_, (hf, hb, hi) in enumerate(hosts):
        frames, boxes, fidx = _frames_and_boxes(F, H, W, per, 100 + s_)
        hf.copy_(t(frames)); hb.copy_(t(boxes)); hi.copy_(t(fidx))
        o = fs.submit(hf, hb, hi)
        outs.append({k_: o[k_].clone() for k_ in ('smpl_vertices', 'smpl_joints2d', 'pred_cam_t', 'cam_vfov')})
    fs.drain()
    assert fs.h2d_bytes == 3 * (F * H * W * 3 + N * 16 + N * 4)
    for s_ in range(3):
        frames, boxes, fidx = _frames_and_boxes(F, H, W, per, 100 + s_)
        k = 0
        for f in range(F):
            n = per[f]
            fr = t(frames[f]).to(dev)
            c = crop_detections(fr, boxes[k:k + n])
            iw, ih = torch.full((n,), float(W), device=dev), torch.full((n,), float(H), device=dev)
            ref = pipe(c['inp_images'], c['bbox_scale'], c['bbox_center'], iw, ih)
            for key in outs[s_]:
                assert torch.equal(outs[s_][key][k:k + n], ref[key]), (s_, f, key)
            k += n


def test_tester_batched_equals_per_frame(tmp_path):
    """SPECTester.run_on_image_folder with detections batched across frames (frame_batch=256 and a small cap that forces
    several flushes) writes the same spec_results pickles as one forward per frame (frame_batch=1)."""
    import joblib
    from types import SimpleNamespace
    from PIL import Image
    from spec_amd import evaluation, synth
    from spec_amd.tester import SPECTester
    d = str(tmp_path)
    evaluation.write_standin_data_tree(d, n_images=1)
    folder = os.path.join(d, 'frames')
    os.makedirs(folder)
    rng = np.random.default_rng(5)
    sizes = [(240, 320), (240, 320), (300, 200), (240, 320), (128, 128), (200, 260)]
    dets = []
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(folder, f'f{i}.png'))
        n = [2, 0, 3, 1, 4, 70][i]        # the last frame alone crosses the kernel-variant thresholds of a small batch
        dets.append(np.stack([rng.uniform(0, w, n), rng.uniform(0, h, n), rng.uniform(40, w, n), rng.uniform(60, h, n)], 1).astype(np.float32))
    hs = {k_: t(v) for k_, v in synth.hmr_state(1002, True).items()}
    cwd = os.getcwd()
    os.chdir(d)
    try:
        results = {}
        for tag, fb, plan in (('per_frame', 1, 'throughput'), ('batched', 256, 'throughput'), ('small_cap', 5, 'throughput'),
                              ('per_frame_default_plan', 1, None), ('batched_default_plan', 256, None), ('per_frame_auto', 1, 'auto')):
            out = os.path.join(d, 'out_' + tag)
            args = SimpleNamespace(cfg=None, ckpt=hs, no_save=False, no_render=True, synthetic_assets=True, frame_batch=fb, plan=plan,
                                   decode_threads=2, camcalib_model=gpu_models(True, True, DEV)[0], detections=dets)
            te = SPECTester(args)
            te.run_camcalib(folder, out)
            n_done = te.run_on_image_folder(folder, te.run_detector(folder), out, None)
            assert n_done == 5
            results[tag] = {f: joblib.load(os.path.join(out, 'spec_results', f)) for f in sorted(os.listdir(os.path.join(out, 'spec_results')))}
    finally:
        os.chdir(cwd)
    assert sorted(results['per_frame']) == ['f0.pkl', 'f2.pkl', 'f3.pkl', 'f4.pkl', 'f5.pkl']
    for tag in ('batched', 'small_cap'):           # one plan: bit-identical whatever the batching (1 ... 80 crops per forward)
        assert sorted(results[tag]) == sorted(results['per_frame'])
        for f, ref in results['per_frame'].items():
            for key, v in ref.items():
                assert results[tag][f][key].shape == v.shape and np.array_equal(results[tag][f][key], v), (tag, f, key)
    # DEFAULT arguments (no --plan): one plan for the whole run whatever --frame_batch, so frame_batch = 1 (the reference's
    # structure, spec/tester.py:143-163: one deterministic result per image) and 256 write identical files
    for tag in ('per_frame_default_plan', 'batched_default_plan'):
        for f, ref in results['per_frame'].items():
            for key, v in ref.items():
                assert np.array_equal(results[tag][f][key], v), (tag, f, key)
    # --plan auto: the single / latency plan by detection count - same results to fp32 rounding
    for f, ref in results['per_frame'].items():
        for key in ('smpl_vertices', 'smpl_joints2d', 'pred_cam_t'):
            a, b = results['per_frame_auto'][f][key].astype(np.float64), ref[key].astype(np.float64)
            assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), (f, key)
    assert results['per_frame']['f4.pkl']['smpl_vertices'].shape == (4, 6890, 3)
    assert results['per_frame']['f5.pkl']['smpl_vertices'].shape == (70, 6890, 3)


def test_concurrent_stream_is_measured():
    """``spec_amd.streams.concurrent_stream``: the helper stream is accepted only if a small upload on it finishes while the
    busy work on the current stream is still running (a stream on the same hardware queue would finish after it)."""
    from spec_amd.streams import concurrent_stream
    a = torch.randn(4096, 4096, device=DEV)
    sink = []

    def busy():
        sink.clear()
        for _ in range(24):
            sink.append(a @ a)

    probe = {}
    st = concurrent_stream(torch.device(DEV), busy, probe=probe)
    assert isinstance(st, torch.cuda.Stream)
    assert probe['step_ms'] > 2.0 and probe['upload_8MB_beside_step_ms'] < 0.5 * probe['step_ms'], probe
    # nothing to hide behind: no measurement, still a stream
    assert isinstance(concurrent_stream(torch.device(DEV), lambda: None, probe={}), torch.cuda.Stream)
