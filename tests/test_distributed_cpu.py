"""N > 1 path on CPU: world_size-2 gloo run of the shard -> forward -> single all-gather
protocol (the GPU forward is replaced by a deterministic per-image stand-in; the collective
and the rank-major packing are the real code from spec_amd.pipeline)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spec_amd.pipeline import (PACKED_KEYS, AsyncGather, gather_outputs, joints_payload, pack_outputs, shard_range,
                               unpack_joints, unpack_outputs)

V = 37  # small synthetic vertex count


def _fake_forward(image_ids):
    """Per-image outputs that depend only on the global image id."""
    out = {}
    for k, shp in PACKED_KEYS:
        shp = (V, 3) if shp is None else shp
        n = int(np.prod(shp)) if len(shp) else 1
        base = torch.arange(n, dtype=torch.float32).reshape(1, *shp) * 1e-3
        out[k] = base + image_ids.to(torch.float32).reshape(-1, *([1] * len(shp)))
    return out


def _worker(rank, world, port, total, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, rank, world)
        out = _fake_forward(torch.arange(lo, hi))
        full = gather_outputs(out)
        # the bench's overlapped variant: 5 "steps" with at most 2 collectives in flight, results in order
        ag = AsyncGather(depth=2, keep_results=True)
        for step in range(5):
            ag.submit({k: v + 100.0 * step for k, v in out.items()})
        steps = ag.drain()
        ok = len(steps) == 5 and all(torch.equal(s_, full + 100.0 * i) for i, s_ in enumerate(steps))
        # the graph-replay flow: records written IN PLACE into two alternating buffers; reserve() before a buffer is
        # overwritten guarantees the collective that still reads it has finished
        rec0 = pack_outputs(out)
        bufs = [torch.empty_like(rec0), torch.empty_like(rec0)]
        ag2 = AsyncGather(depth=2, keep_results=True)
        for step in range(6):
            ag2.reserve()
            buf = bufs[step % 2]
            buf.copy_(rec0 + 100.0 * step)                 # stands for the kernels writing the record
            buf.specmi_static_buffers = 2                  # what GraphedPipeline(buffers=2) marks its records with
            got = ag2.submit({'record': buf, 'pred_cam': out['pred_cam']}, inplace=True)
            assert got.shape[0] == world * rec0.shape[0]
        steps2 = ag2.drain()
        ok = ok and len(steps2) == 6 and all(torch.equal(s_, full + 100.0 * i) for i, s_ in enumerate(steps2))
        # persistent receive buffers: two tensors, reused round-robin
        ok = ok and len(ag2._recv) == 2 and len({t.data_ptr() for t in ag2._recv}) == 2
        # a static record sent WITHOUT inplace is cloned: overwriting it right after submit must not change the result
        ag3 = AsyncGather(depth=2, keep_results=True)
        one = bufs[0]
        one.specmi_static_buffers = 1
        for step in range(4):
            one.copy_(rec0 + 100.0 * step)
            ag3.submit({'record': one, 'pred_cam': out['pred_cam']})
            one.fill_(-1.0)                                # "the next replay" tramples the static buffer
        steps3 = ag3.drain()
        ok = ok and all(torch.equal(s_, full + 100.0 * i) for i, s_ in enumerate(steps3))
        # ... and inplace on it is refused (one buffer < depth 2, or no reserve() before the step)
        for nbuf, do_reserve in ((1, True), (2, False)):
            one.specmi_static_buffers = nbuf
            ag4 = AsyncGather(depth=2)
            if do_reserve:
                ag4.reserve()
            try:
                ag4.submit({'record': one, 'pred_cam': out['pred_cam']}, inplace=True)
                ok = False
            except RuntimeError:
                pass
        # joints-only payload (2,496 B per image)
        agj = AsyncGather(depth=2, keep_results=True, payload='joints')
        agj.submit(out)
        agj.submit({'record': rec0, 'pred_cam': out['pred_cam'], **{k: out[k] for k, _ in PACKED_KEYS}})
        js = agj.drain()
        nj = joints_payload(out).shape[1]
        ok = ok and nj == 624 and all(torch.equal(j, full[:, full.shape[1] - nj:]) for j in js)
        uj = unpack_joints(js[0])
        ok = ok and uj['smpl_joints2d'].shape == (world * rec0.shape[0], 49, 2) and 'smpl_vertices' not in uj
        q.put((rank, full.numpy() if ok else None))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_shard_range_covers_everything():
    for total, world in [(2048, 8), (10, 4), (7, 2), (3, 8)]:
        seen = []
        for r in range(world):
            lo, hi = shard_range(total, r, world)
            seen += list(range(lo, hi))
        assert seen == list(range(total))
    assert shard_range(2048, 3, 8) == (768, 1024)


@pytest.mark.timeout(180)
def test_two_rank_all_gather_matches_unsharded():
    world, total = 2, 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = pack_outputs(_fake_forward(torch.arange(total))).numpy()
    for r in range(world):
        assert np.array_equal(results[r], ref)        # every rank holds the rank-major global result
    back = unpack_outputs(torch.from_numpy(results[0]), V)
    assert back['smpl_vertices'].shape == (total, V, 3)
    assert torch.equal(back['cam_vfov'], torch.arange(total, dtype=torch.float32))
