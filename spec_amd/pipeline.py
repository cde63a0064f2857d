"""The fused per-image path: CamCalib -> soft-argmax decode -> (R, K) -> SPEC -> SMPL -> projection.

The reference runs these as separate processes joined by pickle files
(``spec/tester.py:86-88`` spawns ``scripts/camcalib_demo.py``; ``spec/utils/cam_params.py:28-35``
reads the result back).  Here every hand-off stays in HBM on one stream.

Multi-GPU (``BASELINE.json`` config 4): images shard embarrassingly across ranks (no
cross-image op in eval mode); each rank packs its outputs into one (B, 21294)-float record
and a single RCCL all-gather over xGMI collects them.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import cam_utils

# (key, per-image shape) in packing order: 85,164 B of SPEC outputs + 12 B of camera angles
PACKED_KEYS = (
    ('smpl_vertices', None), ('smpl_joints3d', (49, 3)), ('smpl_joints2d', (49, 2)),
    ('pred_cam_t', (3,)), ('pred_pose', (24, 3, 3)), ('pred_cam', (3,)), ('pred_shape', (10,)),
    ('pred_pose_6d', (144,)), ('cam_vfov', ()), ('cam_pitch', ()), ('cam_roll', ()),
)


class SpecPipeline:
    """``overlap=True`` runs the CamCalib network on a second HIP stream concurrently with the
    SPEC trunk (the two ResNet-50 trunks are independent; only the regressor head needs the
    camera).  On MI355X this fills the partially occupied last wave of workgroups of one
    trunk's kernels with the other trunk's work and lets bandwidth-bound layers of one run
    beside MFMA-bound layers of the other."""

    def __init__(self, camcalib, hmr, overlap: bool = True, packed: bool = True, grouped='auto'):
        """``grouped=True``: when CamCalib sees the same-shaped input as SPEC and both trunks are ResNets of the same depth
        (the benchmark configuration), every trunk layer of BOTH networks is one grouped launch on one stream
        (``specmi_trunk_forward_pair``) instead of two trunks on two streams: half the launches, no stream join - what small
        batches want; results are bit-identical either way.  Falls back to ``overlap`` when the shapes differ.
        ``'auto'`` (default) groups where it measured faster on MI355X (round 5, with the wave-split unit:
        profiles/r05_c_wsplit_check.jsonl, r05_d_structure_sweep.jsonl, r05_l_plan_and_structure_crossover.jsonl): batch 1-3 (0.50 vs
        0.61 ms at batch 1, 0.69 vs 0.72 at 2, 0.88 vs 0.94 at 3) and 17-20 (throughput plan: 2.43 vs 2.60 ms at batch 17, 2.59 vs
        2.83 at 20).  At batch 4-16 two trunks on two streams are ahead - each is then a SINGLE trunk, which keeps the latency plan up
        to 16 images (0.97 vs 1.01 ms at batch 4, 1.44 vs 1.54 at 8, 1.81 vs 2.09 at 11, 2.10 vs 2.22 at 14, 2.40 vs 2.41 at 16: one trunk's launch gaps
        and reduction tails hide under the other's kernels) - and again from 21 (3.00 vs 3.22 ms).
        ``auto_groups(nb)`` is that rule - the ONE place that holds it (bench.py asks ``launch_structure``).
        ``packed=True``: the kernels write every per-image output straight into ONE (B, 21294)-float record (the
        all-gather payload of config 4); the returned tensors are views of it and ``out['record']`` is the record
        itself, so collecting results over RCCL needs no packing copy."""
        self.camcalib = camcalib
        self.hmr = hmr
        self.overlap = overlap
        self.packed = packed
        self.grouped = grouped
        self._side = {}

    @staticmethod
    def auto_groups(nb: int) -> bool:
        """grouped='auto': both trunks per layer as one grouped launch at this batch size?"""
        return nb <= 3 or 17 <= nb <= 20

    def _can_group(self, images_shape, cam_shape) -> bool:
        want_group = self.auto_groups(images_shape[0]) if self.grouped == 'auto' else bool(self.grouped)
        return bool(want_group and self.hmr.use_cam and cam_shape == images_shape and
                    getattr(self.hmr, '_backbone_id', 50) == getattr(self.camcalib, '_backbone_depth', 50) and
                    getattr(self.hmr, 'conv_precision', 0) == 0 and getattr(self.camcalib, 'conv_precision', 0) == 0)

    def launch_structure(self, images_shape, camcalib_shape=None) -> Dict[str, object]:
        """What ``__call__`` does for inputs of these shapes: {'grouped': bool, 'structure': str, 'plan': str} - the launch
        structure (the rule of ``__call__`` itself) and the trunk plan the library reports for it (``specmi_trunk_plan``)."""
        images_shape = tuple(images_shape)
        cam_shape = images_shape if camcalib_shape is None else tuple(camcalib_shape)
        nb = images_shape[0]
        can_group = self._can_group(images_shape, cam_shape)
        if can_group:
            structure = 'both trunks per layer as one grouped launch, one stream'
        elif self.overlap and self.hmr.use_cam:
            structure = 'two trunks on two streams'
        else:
            structure = 'one trunk after the other, one stream'
        plan = None
        eng = getattr(self.camcalib if can_group else self.hmr, '_engine', None)
        if eng is not None:
            plan = eng.trunk_plan(nb, images_shape[2], images_shape[3], pair=can_group)
        return {'grouped': bool(can_group), 'structure': structure, 'plan': plan}

    def _side_stream(self, device, busy=None):
        """The stream CamCalib runs on.  Chosen by measurement when ``busy`` (a callable that enqueues the SPEC trunk on the
        current stream) is given: a stream that shares the main stream's hardware queue would run the two trunks back to back
        (``spec_amd/streams.py``).  Under graph capture, or without ``busy``, the next stream of the pool is taken."""
        if device not in self._side:
            if busy is not None and not torch.cuda.is_current_stream_capturing():
                from .streams import concurrent_stream
                self._side[device] = concurrent_stream(device, busy)
            else:
                self._side[device] = torch.cuda.Stream(device=device)
        return self._side[device]

    @torch.no_grad()
    def __call__(self, images, bbox_scale, bbox_center, img_w, img_h, camcalib_images=None,
                 record: torch.Tensor = None) -> Dict[str, torch.Tensor]:
        """``images``: (B,3,224,224) crops for SPEC.  ``camcalib_images``: what CamCalib sees
        (the full frame in the reference demo; defaults to the same crops, as in the benchmark).
        ``record``: optional caller-owned (B, record_floats) fp32 tensor to write the packed outputs into."""
        cam_in = images if camcalib_images is None else camcalib_images
        device = images.device
        eng = self.hmr.engine(device)
        angles = None
        if record is None and self.packed and self.hmr.use_cam:
            record = torch.empty(images.shape[0], eng.record_layout()[1], device=device, dtype=torch.float32)
        if record is not None:
            v = eng.record_views(record)
            angles = (v['cam_vfov'], v['cam_pitch'], v['cam_roll'])
        can_group = self._can_group(tuple(images.shape), tuple(cam_in.shape))
        if can_group:
            ceng = self.camcalib.engine(device)
            cfeat, feat = ceng.trunk_pair(eng, cam_in, images)
            logits, cam = ceng.camcalib_head_decode(cfeat, img_h=img_h, img_w=img_w, angles_out=angles)   # one launch at small batches
            out = eng.hmr_regress(feat, cam['cam_rotmat'], cam['cam_intrinsics'], bbox_scale, bbox_center, img_w, img_h,
                                  record=record)
        elif not (self.overlap and self.hmr.use_cam):
            logits = self.camcalib(cam_in)
            cam = cam_utils.decode_camera(logits[0], logits[1], logits[2], img_h=img_h, img_w=img_w, angles_out=angles)
            if self.hmr.use_cam:
                out = eng.hmr_forward(images, cam['cam_rotmat'], cam['cam_intrinsics'], bbox_scale, bbox_center,
                                      img_w, img_h, record=record)
            else:
                out = self.hmr(images, cam_rotmat=cam['cam_rotmat'], cam_intrinsics=cam['cam_intrinsics'],
                               bbox_scale=bbox_scale, bbox_center=bbox_center, img_w=img_w, img_h=img_h)
        else:
            main = torch.cuda.current_stream(device)
            side = self._side_stream(device, lambda: eng.trunk(images))
            side.wait_stream(main)                      # inputs were produced on the main stream
            with torch.cuda.stream(side):
                ceng = self.camcalib.engine(device)
                logits, cam = ceng.camcalib_head_decode(ceng.trunk(cam_in), img_h=img_h, img_w=img_w, angles_out=angles)
            feat = eng.trunk(images)                    # SPEC trunk on the main stream, concurrently
            main.wait_stream(side)                      # the head needs (R, K)
            for v in cam.values():
                if v is not None:
                    v.record_stream(main)
            out = eng.hmr_regress(feat, cam['cam_rotmat'], cam['cam_intrinsics'], bbox_scale, bbox_center, img_w, img_h,
                                  record=record)
        out = dict(out)
        out.update({'cam_vfov': cam['vfov'], 'cam_pitch': cam['pitch'], 'cam_roll': cam['roll'],
                    'cam_f_pix': cam['f_pix'], 'cam_rotmat': cam['cam_rotmat'],
                    'cam_intrinsics': cam['cam_intrinsics']})
        if record is not None and self.hmr.use_cam:
            out['record'] = record     # (without use_cam only the angle columns were written: no packed record to hand out)
        return out


class GraphedPipeline:
    """The whole step captured once into a hipGraph (via torch.cuda.CUDAGraph) and replayed: for
    small batches the ~130 kernel launches of a step are launch-bound, a replay costs one
    submission.  Inputs are copied into static buffers; outputs are the static output tensors
    (valid until the next call)."""

    def __init__(self, pipeline: SpecPipeline, images, bbox_scale, bbox_center, img_w, img_h, warmup: int = 2,
                 buffers: int = 1):
        """``buffers`` > 1 captures that many graphs, each writing its own static output set, replayed round-robin:
        a consumer (e.g. the asynchronous all-gather of step s) may still read buffer s % buffers while step s+1
        runs."""
        self.static_in = [t.clone() for t in (images, bbox_scale, bbox_center, img_w, img_h)]
        for _ in range(warmup):                      # allocates workspaces, sets kernel attributes
            pipeline(*self.static_in)
        torch.cuda.synchronize()
        self.graphs, self.static_outs, self.turn = [], [], 0
        for i in range(max(1, buffers)):
            g = torch.cuda.CUDAGraph()
            # thread-local error mode: another thread's runtime calls (e.g. the RCCL watchdog's event queries in a
            # multi-GPU job) must not invalidate this capture
            kw = {'pool': self.graphs[0].pool()} if self.graphs else {}
            with torch.cuda.graph(g, capture_error_mode='thread_local', **kw):
                out = pipeline(*self.static_in)
            self.graphs.append(g)
            if isinstance(out, dict) and out.get('record') is not None:
                out['record'].specmi_static_buffers = max(1, buffers)   # replay overwrites this record: AsyncGather must not send it blindly
            self.static_outs.append(out)
        self.graph, self.static_out = self.graphs[0], self.static_outs[0]

    @torch.no_grad()
    def __call__(self, images, bbox_scale, bbox_center, img_w, img_h):
        for dst, src in zip(self.static_in, (images, bbox_scale, bbox_center, img_w, img_h)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        i = self.turn
        self.turn = (i + 1) % len(self.graphs)
        self.graphs[i].replay()
        return self.static_outs[i]


class DemoPipeline:
    """The path as the reference's demo runs it (``scripts/spec_demo.py``): CamCalib sees the FULL frame at short side 600
    (``camcalib/pano_dataset.py:156-162``, ``scripts/camcalib_demo.py:95-129``) once per frame, SPEC sees the K person crops of
    that frame with the frame's camera (``spec/tester.py:86-88,109-151``).  The reference does this as two processes joined by
    pickle files, one frame and one small batch at a time; here a slab of F equal-sized frames is one step:

        frames (F,H,W,3) uint8 in HBM --specmi_resize_normalize x F--> (F,3,600,W') --CamCalib, batch F--> decode -> R, K (F)
                                      +--specmi_crop_normalize_batch--> (N,3,224,224) --SPEC trunk--> head(R[frame], K[frame]) -> SMPL

    CamCalib runs on a second stream beside the SPEC trunk (both are MFMA-bound; the join is the regressor head, which needs the
    camera).  Everything is enqueued without a host synchronisation, so the step can be captured into a hipGraph."""

    def __init__(self, camcalib, hmr, min_size: int = 600, crop_size: int = 224, overlap: bool = True):
        self.camcalib, self.hmr, self.min_size, self.crop_size, self.overlap = camcalib, hmr, int(min_size), int(crop_size), overlap
        self._side = {}

    @torch.no_grad()
    def __call__(self, frames_u8, boxes, frame_index, record: torch.Tensor = None) -> Dict[str, torch.Tensor]:
        """frames_u8 (F,H,W,3) uint8 device slab; boxes (N,4) [cx, cy, w, h] device; frame_index (N,) int32 device."""
        from .preprocess import camcalib_transform_batch, crop_detections_batch
        device = frames_u8.device
        F, H, W = frames_u8.shape[:3]
        N = boxes.shape[0]
        eng = self.hmr.engine(device)
        angles = None
        fh = torch.full((F,), float(H), device=device)
        fw = torch.full((F,), float(W), device=device)

        def cam_side():
            cam_in = camcalib_transform_batch(frames_u8, self.min_size)
            logits = self.camcalib(cam_in)
            return cam_utils.decode_camera(logits[0], logits[1], logits[2], img_h=fh, img_w=fw)

        main = torch.cuda.current_stream(device)
        if self.overlap:
            if device not in self._side:
                self._side[device] = torch.cuda.Stream(device=device)
            side = self._side[device]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                cam = cam_side()
        crops = crop_detections_batch(frames_u8, frame_index, boxes, scale=1.0, crop_size=self.crop_size)
        feat = eng.trunk(crops['inp_images'])
        if self.overlap:
            main.wait_stream(side)
            for v in cam.values():
                if v is not None:
                    v.record_stream(main)
        else:
            cam = cam_side()
        fi = frame_index.long()
        R, K = cam['cam_rotmat'].index_select(0, fi), cam['cam_intrinsics'].index_select(0, fi)   # the frame's camera for each of its crops
        img_w = torch.full((N,), float(W), device=device)
        img_h = torch.full((N,), float(H), device=device)
        out = dict(eng.hmr_regress(feat, R, K, crops['bbox_scale'], crops['bbox_center'], img_w, img_h, record=record))
        out.update({'cam_vfov': cam['vfov'], 'cam_pitch': cam['pitch'], 'cam_roll': cam['roll'], 'cam_f_pix': cam['f_pix'],
                    'cam_rotmat': cam['cam_rotmat'], 'cam_intrinsics': cam['cam_intrinsics'], 'bbox_scale': crops['bbox_scale'],
                    'bbox_center': crops['bbox_center']})
        return out


class GraphedStep:
    """Any no-host-sync step ``fn(*tensors) -> dict`` captured once into a hipGraph and replayed; inputs are copied into static
    buffers unless they already are those buffers (``static_in``), outputs are the static output tensors."""

    def __init__(self, fn, *inputs, warmup: int = 2):
        self.static_in = [t.clone() for t in inputs]
        for _ in range(warmup):
            fn(*self.static_in)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.static_out = fn(*self.static_in)

    @torch.no_grad()
    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.static_out


def shard_range(total: int, rank: int, world: int):
    """Contiguous rank-major image slice [lo, hi) of a global batch (config 4: 2048 -> 256 per GPU).
    Remainders go to the lowest ranks, so any total / world is covered exactly once."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_outputs(out: Dict[str, torch.Tensor]) -> torch.Tensor:
    """The (B, record) tensor of a step: the record the kernels wrote (``SpecPipeline(packed=True)``, no copy), or a
    concatenation when the outputs are separate tensors."""
    rec = out.get('record')
    if rec is not None and rec.is_contiguous():
        return rec
    B = out['pred_cam'].shape[0]
    return torch.cat([out[k].reshape(B, -1) for k, _ in PACKED_KEYS], dim=1).contiguous()


def unpack_outputs(packed: torch.Tensor, num_verts: int) -> Dict[str, torch.Tensor]:
    B = packed.shape[0]
    res, off = {}, 0
    for k, shp in PACKED_KEYS:
        shp = (num_verts, 3) if shp is None else shp
        n = 1
        for s in shp:
            n *= s
        res[k] = packed[:, off:off + n].reshape(B, *shp)
        off += n
    return res


JOINT_KEYS = tuple(k for k, _ in PACKED_KEYS if k != 'smpl_vertices')


def joints_payload(out: Dict[str, torch.Tensor]) -> torch.Tensor:
    """The record without its vertices (joints3d | joints2d | cam_t | pose | cam | shape | pose6d |
    camera angles = 624 floats = 2,496 B per image) as a fresh contiguous (B, 624) tensor: 0.64 MB per rank at B = 256
    instead of 21.8 MB.  Always a copy (the columns are a strided slice of the record)."""
    rec = out.get('record')
    B = out['pred_cam'].shape[0]
    if rec is not None:
        nj = sum(out[k].reshape(B, -1).shape[1] for k in JOINT_KEYS)
        return rec[:, rec.shape[1] - nj:].contiguous()
    return torch.cat([out[k].reshape(B, -1) for k in JOINT_KEYS], dim=1).contiguous()


def unpack_joints(packed: torch.Tensor) -> Dict[str, torch.Tensor]:
    B = packed.shape[0]
    res, off = {}, 0
    for k, shp in PACKED_KEYS:
        if shp is None:
            continue
        n = 1
        for s_ in shp:
            n *= s_
        res[k] = packed[:, off:off + n].reshape(B, *shp)
        off += n
    return res


class AsyncGather:
    """The per-step all-gather taken off the critical path: ``submit(out)`` starts the collective with
    ``async_op=True`` (RCCL runs it on its own stream once the step's kernels are done), so it overlaps the NEXT
    step's kernels; at most ``depth`` collectives are in flight (the oldest is waited for before a new one is
    queued), ``drain()`` waits for the rest.  Results come back in submission order.

    Buffers: the receive side is ``depth`` PERSISTENT tensors used round-robin (no per-step allocation: 174 MB at
    N = 8); a result is valid until ``depth`` further submits.  The send side is safe by default:

    * a record that is a fresh tensor of this step (``SpecPipeline`` called eagerly) is sent as it is and kept
      referenced until its collective has finished;
    * a STATIC record that the next graph replay overwrites (``GraphedPipeline`` marks its records with
      ``specmi_static_buffers`` = number of alternating buffers) is CLONED first, unless the caller asks for
      ``submit(out, inplace=True)`` - which is only accepted when the pipeline alternates at least ``depth`` buffers
      and ``reserve()`` was called before the step was enqueued (so the collective that last read the buffer the
      step wrote had finished); anything else raises instead of silently corrupting gathered results;
    * ``payload='joints'`` sends the 2.5 KB-per-image record without vertices (always a small fresh copy).
    """

    def __init__(self, depth: int = 2, group=None, keep_results: bool = False, payload: str = 'full'):
        if payload not in ('full', 'joints'):
            raise ValueError("payload must be 'full' or 'joints'")
        self.depth, self.group, self.keep, self.payload = max(1, depth), group, keep_results, payload
        self.pending = []
        self.results = []
        self._recv = []          # persistent receive buffers
        self._turn = 0
        self._reserved = False   # reserve() called since the last submit

    def _retire(self):
        work, full, _send = self.pending.pop(0)
        work.wait()
        if self.keep:
            self.results.append(full.clone())     # the receive buffer itself is reused
        self.last = full

    def reserve(self):
        """Call BEFORE enqueueing a step whose output buffer may be one a pending collective still reads
        (``GraphedPipeline(buffers=depth)`` reuses buffer s % depth): retires the oldest collectives until a slot is
        free.  ``work.wait()`` makes the current stream wait, not the host."""
        while len(self.pending) >= self.depth:
            self._retire()
        self._reserved = True

    def _recv_buffer(self, rows, cols, like):
        if len(self._recv) < self.depth or any(t.shape != (rows, cols) or t.device != like.device or t.dtype != like.dtype
                                               for t in self._recv):
            if self.pending:          # shape change with collectives in flight: finish them before dropping their buffers
                self.drain()
            self._recv = [torch.empty(rows, cols, device=like.device, dtype=like.dtype) for _ in range(self.depth)]
            self._turn = 0
        buf = self._recv[self._turn]
        self._turn = (self._turn + 1) % self.depth
        return buf

    def submit(self, out: Dict[str, torch.Tensor], inplace: bool = False):
        import torch.distributed as dist
        reserved = self._reserved
        self.reserve()
        self._reserved = False
        if self.payload == 'joints':
            send = joints_payload(out)
        else:
            send = pack_outputs(out)
            static = getattr(send, 'specmi_static_buffers', None)
            if static is not None:
                if not inplace:
                    send = send.clone()
                elif int(static) < self.depth or not reserved:
                    raise RuntimeError(
                        f'AsyncGather.submit(inplace=True): the record is a static graph buffer ({int(static)} alternating '
                        f'buffers, depth {self.depth}, reserve() before the step: {reserved}); the next replay would '
                        'overwrite it while the collective still reads it.  Use GraphedPipeline(buffers >= depth) and call '
                        'reserve() before every step, or submit without inplace.')
        world = dist.get_world_size(self.group)
        full = self._recv_buffer(world * send.shape[0], send.shape[1], send)
        work = dist.all_gather_into_tensor(full, send, group=self.group, async_op=True)
        self.pending.append((work, full, send))     # send stays referenced until the collective has finished
        return full

    def drain(self):
        while self.pending:
            self._retire()
        return self.results if self.keep else getattr(self, 'last', None)


def gather_outputs(out: Dict[str, torch.Tensor], group=None) -> torch.Tensor:
    """One all-gather (RCCL over xGMI with backend 'nccl'; gloo on CPU tests) of the packed
    per-image records: returns (world*B, record) on every rank, rank-major."""
    import torch.distributed as dist
    packed = pack_outputs(out)
    world = dist.get_world_size(group)
    full = torch.empty(world * packed.shape[0], packed.shape[1], device=packed.device, dtype=packed.dtype)
    dist.all_gather_into_tensor(full, packed, group=group)
    return full
