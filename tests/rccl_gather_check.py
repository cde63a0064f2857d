#!/usr/bin/env python
"""Real multi-rank check of the N > 1 path (run by tests/test_gpu_round2.py when >= 2 GPUs are visible, or by hand:
``python tests/rccl_gather_check.py --gpus 2``): launches one process per GPU under torch.distributed.run (RCCL),
every rank runs its contiguous shard of a global batch through the fused pipeline - records written in place by
the kernels, hipGraph replay with two alternating record buffers, asynchronous all-gather - and rank 0 compares
the gathered (world*B, 21294) records bit for bit with the unsharded forward of the whole batch."""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=2)
    ap.add_argument('--per-rank', type=int, default=6)
    args = ap.parse_args()
    if 'WORLD_SIZE' not in os.environ:
        import torch
        if torch.cuda.device_count() < args.gpus:
            print(f'need {args.gpus} GPUs, {torch.cuda.device_count()} visible', file=sys.stderr)
            sys.exit(2)
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4')
        sys.exit(subprocess.call([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                  f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:], env=env))

    import torch
    import torch.distributed as dist
    from spec_amd import synth
    from spec_amd.pipeline import SpecPipeline, GraphedPipeline, AsyncGather, gather_outputs, shard_range
    from tests.util import gpu_models, t
    torch.set_grad_enabled(False)
    world, rank, local = int(os.environ['WORLD_SIZE']), int(os.environ['RANK']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist.init_process_group('nccl', device_id=dev)
    assert dist.get_world_size() == args.gpus == world
    cc, hm = gpu_models(True, True, dev)
    pipe = SpecPipeline(cc, hm, overlap=True)
    total = args.per_rank * world
    steps = 4
    batches = []
    for s_ in range(steps):
        x = t(synth.images(50 + s_, total))
        sc, ce, iw, ih = [t(a) for a in synth.bbox_inputs(50 + s_, total, 640., 480.)]
        batches.append((x, sc, ce, iw, ih))
    lo, hi = shard_range(total, rank, world)
    mine = [[a[lo:hi].contiguous().to(dev) for a in b] for b in batches]
    # (1) blocking gather of an eager step
    full0 = gather_outputs(pipe(*mine[0]))
    # (2) the bench's flow: graph replay, two record buffers, asynchronous gathers
    gp = GraphedPipeline(pipe, *mine[0], buffers=2)
    ag = AsyncGather(depth=2, keep_results=True)
    for s_ in range(steps):
        ag.reserve()
        ag.submit(gp(*mine[s_]), inplace=True)
    res = ag.drain()
    # (3) ONE static record buffer and no reserve(): the default submit() clones, so the next replay cannot corrupt the
    # collective; plus the joints-only payload (record without vertices)
    gp1 = GraphedPipeline(pipe, *mine[0], buffers=1)
    ag1 = AsyncGather(depth=2, keep_results=True)
    agj = AsyncGather(depth=2, keep_results=True, payload='joints')
    for s_ in range(steps):
        o = gp1(*mine[s_])
        ag1.submit(o)
        agj.submit(o)
    res1, resj = ag1.drain(), agj.drain()
    refused = False
    try:
        ag1.submit(gp1(*mine[0]), inplace=True)
    except RuntimeError:
        refused = True
    ag1.drain()
    torch.cuda.synchronize()
    ok = refused
    if not refused:
        print('inplace submit of a single static buffer was not refused')
    if rank == 0:
        for s_ in range(steps):
            ref = pipe(*[a.to(dev) for a in batches[s_]])['record']
            if not torch.equal(res[s_], ref):
                ok = False
                print('step', s_, 'mismatch: max abs diff', float((res[s_] - ref).abs().max()))
            if not torch.equal(res1[s_], ref) or not torch.equal(resj[s_], ref[:, ref.shape[1] - 624:]):
                ok = False
                print('step', s_, 'cloned / joints-only gather mismatch')
        if not torch.equal(full0, res[0]):
            ok = False
            print('blocking gather differs from the asynchronous one')
    flag = torch.tensor([1 if ok else 0], device=dev)
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and ok:
        print(f'RCCL_GATHER_OK world={world} records={tuple(res[0].shape)} backend=nccl', flush=True)
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == '__main__':
    main()
