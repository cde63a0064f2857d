"""Real-input front end of the path: uint8 video frames in HOST memory -> crops -> the same step the benchmark times.

The reference walks an image folder one frame at a time (``spec/tester.py:109-151``): decode, one crop per detection on
the CPU, a blocking copy to the GPU, one forward with batch = #detections of that frame.  At 8 people per frame that is a
batch of 8 per launch set - the small-batch regime (2 ms per step, 4 k images/s) instead of the 256-image step (29 ms, 8.8 k
images/s).  ``FrameStream`` keeps the full-size step fed from host frames:

* a frame slab (F equal-sized frames) and its detections live in PINNED host memory (a decoder writes there);
* two device slabs alternate: a copy stream uploads slab s + 1 (H2D over PCIe, ~57 GB/s) while the compute stream still
  works on slab s;
* ONE batched crop launch (``specmi_crop_normalize_batch``) cuts all detections of the F frames straight into the static
  input buffers of the step (no intermediate copy), then the step runs (eagerly or as a hipGraph replay);
* events order the two streams: the upload of a slab waits for the crop launch that last read it, the crop launch waits
  for its upload.  Nothing blocks the host except ``drain()``.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .preprocess import crop_detections_batch


class FrameStream:
    def __init__(self, step: Callable, device, frame_hw, frames_per_step: int, crops_per_step: int, crop_size: int = 224,
                 slots: int = 2, scale: float = 1.0, copy_stream: Optional[torch.cuda.Stream] = None):
        """``step(images, bbox_scale, bbox_center, img_w, img_h)`` is a ``SpecPipeline`` or ``GraphedPipeline``.  When it
        has ``static_in`` (a captured graph) the crops are written directly into those buffers.  ``copy_stream``: the
        stream of the uploads; by default one is chosen by measurement (``_pick_copy_stream``)."""
        self.step, self.device = step, torch.device(device)
        self.H, self.W = int(frame_hw[0]), int(frame_hw[1])
        self.F, self.N, self.S, self.scale = int(frames_per_step), int(crops_per_step), int(crop_size), float(scale)
        dev = self.device
        self.slabs = [torch.empty(self.F, self.H, self.W, 3, dtype=torch.uint8, device=dev) for _ in range(slots)]
        self.boxes = [torch.empty(self.N, 4, dtype=torch.float32, device=dev) for _ in range(slots)]
        self.fidx = [torch.empty(self.N, dtype=torch.int32, device=dev) for _ in range(slots)]
        static = getattr(step, 'static_in', None)
        if static is not None:
            if static[0].shape != (self.N, 3, self.S, self.S):
                raise ValueError(f'the captured step takes {tuple(static[0].shape)} crops, FrameStream was asked for '
                                 f'{(self.N, 3, self.S, self.S)}')
            self.x, self.sc, self.ce, self.img_w, self.img_h = static
        else:
            self.x = torch.empty(self.N, 3, self.S, self.S, dtype=torch.float32, device=dev)
            self.sc = torch.empty(self.N, dtype=torch.float32, device=dev)
            self.ce = torch.empty(self.N, 2, dtype=torch.float32, device=dev)
            self.img_w = torch.empty(self.N, dtype=torch.float32, device=dev)
            self.img_h = torch.empty(self.N, dtype=torch.float32, device=dev)
        self.img_w.fill_(float(self.W))
        self.img_h.fill_(float(self.H))
        if static is None:                 # benign inputs for the probe step below (a captured step holds its capture inputs)
            self.x.zero_(); self.sc.fill_(1.0); self.ce.zero_()
        self.copy_stream = copy_stream if copy_stream is not None else self._pick_copy_stream()
        self.ready = [torch.cuda.Event() for _ in range(slots)]
        self.free = [None] * slots        # recorded after the crop launch that read the slot
        self.turn = 0
        self.h2d_bytes = 0

    def _pick_copy_stream(self, tries: int = 6):
        """A copy stream whose uploads really run beside the step (``spec_amd/streams.py``): a stream that shares the hardware
        queue of the stream the step is launched on gets its upload dispatched behind the step's ~120 kernels - measured 33.5
        instead of 29.6 ms per step with 1080p slabs (``scripts/e2e_probe.py``: the same code gave 0.98 or 0.86 of the
        HBM-resident rate depending on how many streams the process had created before).  At most ``tries`` steps at
        construction.  Default priority: a high-priority stream measured lower (0.91-0.92 against 0.97-0.98)."""
        from .streams import concurrent_stream
        args = (self.x, self.sc, self.ce, self.img_w, self.img_h)
        self.copy_probe = {}
        return concurrent_stream(self.device, lambda: self.step(*args), tries=tries, probe=self.copy_probe)

    def host_buffers(self):
        """Pinned host staging for one step: (frames (F,H,W,3) uint8, boxes (N,4) fp32, frame_index (N,) int32)."""
        return (torch.empty(self.F, self.H, self.W, 3, dtype=torch.uint8).pin_memory(),
                torch.empty(self.N, 4, dtype=torch.float32).pin_memory(),
                torch.empty(self.N, dtype=torch.int32).pin_memory())

    @torch.no_grad()
    def submit(self, frames_host: torch.Tensor, boxes_host: torch.Tensor, fidx_host: torch.Tensor):
        """Enqueue one step on frames that sit in (pinned) host memory.  Returns the step's outputs (device tensors; with a
        captured graph: its static outputs, valid until the next submit).  The host buffers may be refilled once
        ``uploaded()`` / ``drain()`` says the copy has finished - or simply use one pinned set per slot."""
        if fidx_host.numel() and (int(fidx_host.min()) < 0 or int(fidx_host.max()) >= self.F):
            raise ValueError(f'frame_index values must lie in [0, {self.F})')     # checked while the index is on the host
        i = self.turn
        self.turn = (i + 1) % len(self.slabs)
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.copy_stream):
            if self.free[i] is not None:
                self.copy_stream.wait_event(self.free[i])          # the crop launch that last read this slab is done
            self.slabs[i].copy_(frames_host, non_blocking=True)
            self.boxes[i].copy_(boxes_host, non_blocking=True)
            self.fidx[i].copy_(fidx_host, non_blocking=True)
            self.ready[i].record(self.copy_stream)
        self.h2d_bytes += frames_host.numel() + boxes_host.numel() * 4 + fidx_host.numel() * 4
        main.wait_event(self.ready[i])
        crop_detections_batch(self.slabs[i], self.fidx[i], self.boxes[i], scale=self.scale, crop_size=self.S,
                              out={'inp_images': self.x, 'bbox_scale': self.sc, 'bbox_center': self.ce})
        ev = torch.cuda.Event()
        ev.record(main)
        self.free[i] = ev
        return self.step(self.x, self.sc, self.ce, self.img_w, self.img_h)

    def uploaded(self, slot: int = None, wait: bool = False) -> bool:
        """Has the host->device copy of the step submitted into ``slot`` (default: the most recent submit) finished, i.e.
        may its host buffers be refilled?  ``wait=True`` blocks until it has."""
        i = (self.turn - 1) % len(self.slabs) if slot is None else slot
        if wait:
            self.ready[i].synchronize()
            return True
        return bool(self.ready[i].query())

    def drain(self):
        torch.cuda.synchronize(self.device)
