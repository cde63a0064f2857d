"""bench.py's N > 1 control flow, end to end, without hardware: ``--backend gloo --fake-forward`` runs the same
main() - self_launch under torch.distributed.run, the process group, the 16-image probe (ragged at world 3: 16 % 3 != 0 takes the pad
path), two alternating record buffers + AsyncGather with in-place sends, the per-rank timing gather, the ``comm`` block, the CPU
baseline on rank 0 - with the GPU forward replaced by a per-image stand-in.  The first time ≥ 2 RCCL ranks run on hardware is then
not also the first time this code runs.  (one all-gather of the packed records per step.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, extra=()):
    env = dict(os.environ, OMP_NUM_THREADS='2', HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None); env.pop('MASTER_PORT', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--backend', 'gloo', '--fake-forward', '--steps', '4',
           '--warmup', '1', '--batch', '6', '--sustained-seconds', '0.2', '--cpu-baseline-seconds', '1.5', *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]           # ONE JSON line, from rank 0
    return json.loads(lines[0]), r.stderr


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world', [2, 3])
def test_bench_main_runs_the_multi_rank_control_flow(world):
    line, err = _run(world)
    assert line['dry_run'] is True and line['n_gpus'] == world and line['steps'] == 4 and line['warmup'] == 1
    assert line['metric'].startswith('images/sec') and line['scaling'] == 'weak' and line['value'] > 0
    assert line['config']['global_batch'] == 6 * world and line['config']['launch'] == 'hipGraph replay'
    comm = line['comm']
    assert comm['backend'] == 'gloo' and comm['world_size'] == world
    assert comm['probe_gathered_16_images_equal_unsharded'] is True
    assert comm['record_bytes_per_image'] == 21294 * 4
    assert len(comm['per_rank_images_per_s']) == world and all(v > 0 for v in comm['per_rank_images_per_s'])
    assert comm['ms_per_step_with_async_gather'] > 0 and comm['ms_per_step_without_gather'] > 0
    assert line['sustained']['steps'] >= 4
    cpu = line['cpu_baseline']                                  # rank 0 times the oracle when N > 1 too
    assert cpu and cpu['kind'] == 'port' and cpu['value'] > 0 and cpu['cores'] >= 1
    assert cpu['ranks_waiting_at_the_barrier_meanwhile'] == world - 1
    # the compact scalars are the LAST line of stderr
    last = [l for l in err.strip().splitlines() if l.strip()][-1]
    assert last.startswith('[bench] summary {'), last
    summ = json.loads(last[len('[bench] summary '):])
    assert summ['value_images_per_s'] == line['value'] and summ['comm_world_size'] == world
    assert summ['comm_probe_equal_unsharded'] is True and len(last) < 1600


@pytest.mark.timeout(600)
def test_bench_joints_payload_and_eager_steps():
    """The 2,496-byte payload (fresh copies, no in-place send) and eager launches (no alternating buffers)."""
    line, _ = _run(2, ('--gather', 'joints', '--no-graph', '--no-cpu-baseline', '--no-sustained'))
    assert line['comm']['payload'] == 'joints' and line['comm']['sent_bytes_per_image'] == 2496
    assert line['config']['launch'] == 'eager launches' and line['cpu_baseline'] is None
    assert line['comm']['probe_gathered_16_images_equal_unsharded'] is True


def test_bench_refuses_mixed_modes():
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    for extra in (['--fake-forward'], ['--backend', 'gloo']):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *extra], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert r.returncode == 2 and 'go together' in r.stderr
