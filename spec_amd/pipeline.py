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

    def __init__(self, camcalib, hmr, overlap: bool = True):
        self.camcalib = camcalib
        self.hmr = hmr
        self.overlap = overlap
        self._side = {}

    def _side_stream(self, device):
        if device not in self._side:
            self._side[device] = torch.cuda.Stream(device=device)
        return self._side[device]

    @torch.no_grad()
    def __call__(self, images, bbox_scale, bbox_center, img_w, img_h, camcalib_images=None) -> Dict[str, torch.Tensor]:
        """``images``: (B,3,224,224) crops for SPEC.  ``camcalib_images``: what CamCalib sees
        (the full frame in the reference demo; defaults to the same crops, as in the benchmark)."""
        cam_in = images if camcalib_images is None else camcalib_images
        if not (self.overlap and self.hmr.use_cam):
            logits = self.camcalib(cam_in)
            cam = cam_utils.decode_camera(logits[0], logits[1], logits[2], img_h=img_h, img_w=img_w)
            out = self.hmr(images, cam_rotmat=cam['cam_rotmat'], cam_intrinsics=cam['cam_intrinsics'],
                           bbox_scale=bbox_scale, bbox_center=bbox_center, img_w=img_w, img_h=img_h)
        else:
            device = images.device
            main = torch.cuda.current_stream(device)
            side = self._side_stream(device)
            side.wait_stream(main)                      # inputs were produced on the main stream
            with torch.cuda.stream(side):
                logits = self.camcalib(cam_in)
                cam = cam_utils.decode_camera(logits[0], logits[1], logits[2], img_h=img_h, img_w=img_w)
            eng = self.hmr.engine(device)
            feat = eng.trunk(images)                    # SPEC trunk on the main stream, concurrently
            main.wait_stream(side)                      # the head needs (R, K)
            for v in cam.values():
                if v is not None:
                    v.record_stream(main)
            out = eng.hmr_head(feat, cam['cam_rotmat'], cam['cam_intrinsics'], img_h)
            out.update(eng.smpl(out['pred_pose'], out['pred_shape'], out['pred_cam'], cam['cam_rotmat'],
                                cam['cam_intrinsics'], bbox_scale, bbox_center, img_w, img_h))
        out.update({'cam_vfov': cam['vfov'], 'cam_pitch': cam['pitch'], 'cam_roll': cam['roll'],
                    'cam_f_pix': cam['f_pix'], 'cam_rotmat': cam['cam_rotmat'],
                    'cam_intrinsics': cam['cam_intrinsics']})
        return out


class GraphedPipeline:
    """The whole step captured once into a hipGraph (via torch.cuda.CUDAGraph) and replayed: for
    small batches the ~130 kernel launches of a step are launch-bound, a replay costs one
    submission.  Inputs are copied into static buffers; outputs are the static output tensors
    (valid until the next call)."""

    def __init__(self, pipeline: SpecPipeline, images, bbox_scale, bbox_center, img_w, img_h, warmup: int = 2):
        self.static_in = [t.clone() for t in (images, bbox_scale, bbox_center, img_w, img_h)]
        for _ in range(warmup):                      # allocates workspaces, sets kernel attributes
            pipeline(*self.static_in)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread-local error mode: another thread's runtime calls (e.g. the RCCL watchdog's event queries in a
        # multi-GPU job) must not invalidate this capture
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.static_out = pipeline(*self.static_in)

    @torch.no_grad()
    def __call__(self, images, bbox_scale, bbox_center, img_w, img_h):
        for dst, src in zip(self.static_in, (images, bbox_scale, bbox_center, img_w, img_h)):
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


class AsyncGather:
    """The per-step all-gather taken off the critical path: ``submit(out)`` packs the step's outputs and starts
    the collective with ``async_op=True`` (RCCL runs it on its own stream once the packing kernel is done), so it
    overlaps the NEXT step's kernels; at most ``depth`` collectives are in flight (the oldest is waited for before
    a new one is queued, which also bounds memory), ``drain()`` waits for the rest.  Results come back in
    submission order."""

    def __init__(self, depth: int = 2, group=None, keep_results: bool = False):
        self.depth, self.group, self.keep = max(1, depth), group, keep_results
        self.pending = []
        self.results = []

    def _retire(self):
        work, full = self.pending.pop(0)
        work.wait()
        if self.keep:
            self.results.append(full)
        self.last = full

    def submit(self, out: Dict[str, torch.Tensor]):
        import torch.distributed as dist
        while len(self.pending) >= self.depth:
            self._retire()
        packed = pack_outputs(out)
        world = dist.get_world_size(self.group)
        full = torch.empty(world * packed.shape[0], packed.shape[1], device=packed.device, dtype=packed.dtype)
        work = dist.all_gather_into_tensor(full, packed, group=self.group, async_op=True)
        self.pending.append((work, full))
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
