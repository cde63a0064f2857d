"""Streams that really run beside the stream the work is launched on.

HIP multiplexes a process's streams onto a few hardware queues, and a hardware queue hands its packets out in order.  A second
stream that lands on the SAME queue as the main stream therefore does not overlap with it at all: its copy / kernel is
dispatched behind everything the main stream has queued.  Which queue a new stream gets depends on how many streams the process
created before (measured on MI355X, `profiles/r03_t_copy_stream_queue_aliasing.txt`: an 8 MB upload took 0.18 ms beside the
29 ms step on one stream and 29 ms on the next two).  So a helper stream is chosen by measurement, once, when it is created.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch


def concurrent_stream(device, launch_busy: Callable[[], None], tries: int = 6, min_busy_s: float = 2e-3, probe: Optional[dict] = None):
    """A new stream on `device` whose work overlaps with what ``launch_busy()`` enqueues on the CURRENT stream.

    ``launch_busy`` must enqueue (not wait for) a few milliseconds of GPU work, e.g. one step of the pipeline.  For each
    candidate stream: start the busy work, put a small pinned-host upload on the candidate and time it on the host; a
    candidate that shares the busy stream's hardware queue finishes only when the busy work does.  Returns the first candidate
    that finishes within a quarter of the busy time (else the best of ``tries``).  When the busy work is shorter than
    ``min_busy_s`` it is repeated until the window is that long."""
    dev = torch.device(device)
    with torch.no_grad():
        launch_busy()                                      # warm: plans, workspaces, kernel attributes
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        launch_busy()
        torch.cuda.synchronize(dev)
        t_busy = time.perf_counter() - t0
        if t_busy < min_busy_s:
            # a short step (small batches: 0.3-0.8 ms) is repeated until the busy window is long enough to measure against - an
            # unmeasured stream that shares the main stream's hardware queue would serialise the two trunks (seen: 1.96 instead
            # of 1.44 ms for the batch-8 step)
            reps = min(64, int(min_busy_s / max(t_busy, 1e-5)) + 1)
            one = launch_busy

            def launch_busy():
                for _ in range(reps):
                    one()
            t_busy *= reps
        src = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
        dst = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
        best, seen = None, []
        for _ in range(tries):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                dst.copy_(src, non_blocking=True)           # first use of the stream, outside the measurement
            torch.cuda.synchronize(dev)
            launch_busy()                                  # asynchronous: the kernels are now queued / running
            t0 = time.perf_counter()
            ev = torch.cuda.Event()
            with torch.cuda.stream(st):
                dst.copy_(src, non_blocking=True)
                ev.record(st)
            ev.synchronize()
            dt = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            seen.append(round(dt * 1e3, 3))
            if best is None or dt < best[0]:
                best = (dt, st)
            if dt < 0.25 * t_busy:
                break
        if probe is not None:
            probe.update({'step_ms': round(t_busy * 1e3, 3), 'upload_8MB_beside_step_ms': round(best[0] * 1e3, 3),
                          'candidates_ms': seen})
        return best[1]
