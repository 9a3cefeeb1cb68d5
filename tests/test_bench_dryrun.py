"""Host side of bench.py at world size 2 (the launcher of the benchmark contract, one process per "GPU") with the device
calls answered by the oracle (tests/dryrun_bench.py): worker threads, steady-state timing, one gather per round of steps
on the main thread, barriers, the maximum over ranks and ONE JSON line from rank 0.  The kernels are not involved; this
guards the N > 1 control flow that cannot be run on the single-GPU test box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(nproc, steps, port, extra=(), launcher=False, size=('--size', '192')):
    """`launcher`: ranks started by torch.distributed.run (what the driver does); else `--gpus N` alone, and bench.py starts
    its own ranks"""
    tail = [os.path.join(ROOT, 'tests', 'dryrun_bench.py'), '--gpus', str(nproc)] + list(size) + ['--steps', str(steps), '--warmup', '1'] + list(extra)
    if launcher:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
               '--master-port', str(port)] + tail
    else:
        cmd = [sys.executable] + tail
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line from rank 0, got %d' % len(lines)
    return json.loads(lines[0])


def test_bench_control_flow_two_ranks_under_the_launcher():
    import pytest
    pytest.importorskip('torch')
    d = run_bench(2, 9, 29631, launcher=True)
    assert d['n_gpus'] == 2 and d['steps'] == 9 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['higher_is_better'] is True and d['vs_baseline'] is None and d['value'] > 0
    assert abs(d['value'] - 2 * 9 * 192 * 192 / (d['ms_per_step'] * 9 / 1e3) / 1e6) / d['value'] < 1e-3      # whole-job aggregate
    assert set(d['roofline']) >= {'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'}
    assert d['config']['images_in_flight_per_gpu'] == 4 and 'gathered in rank 0' in d['config']['parallelism']
    assert d['ms_per_step_incl_fill_drain'] > 0 and 'steady state' in d['config']['timing']
    assert 'cpu_baseline' not in d                     # rank 0 at N = 1 only
    assert len(d['per_rank']['value']) == 2 and all(v > 0 for v in d['per_rank']['value'])      # a slow rank is visible in the line
    assert len(d['per_rank']['host_link_gb_per_s']) == 2 and len(d['per_rank']['placement']) == 2
    # input staging at N > 1: measured on every rank, all ranks at once (either form may win here; --input-ring 0 / 1 fixes it)
    assert d['config']['input_memory'].startswith('pageable numpy') and 'probe on this rank with all 2 ranks' in d['config']['input_ring']
    assert len(d['per_rank']['input_staging']) == 2 and len(d['per_rank']['host_memory_traffic_gb_per_s']) == 2
    assert d['gathered_maps_checked'] == 2 and d['gathered_maps_equal_senders_own'] is True      # what rank 0 received = what each rank holds


def test_bench_batch_config_two_self_spawned_ranks():
    """`--gpus 2` without a launcher: bench.py starts its own ranks; config 4 with the group model of the reference's run, and
    rank 0 checks the CRC of EVERY gathered label map against that run (the oracle stands in for the kernels here, and it
    reproduces the reference's maps bit for bit)"""
    d = run_bench(2, 3, 29633, extra=['--config', '4', '--input-ring', '1'], size=())
    assert d['n_gpus'] == 2 and d['config']['bench_config'] == 4 and d['config']['images_per_step_per_gpu'] == 8
    assert abs(d['value'] - 2 * 3 * 8 * 647 * 1024 / (d['ms_per_step'] * 3 / 1e3) / 1e6) / d['value'] < 1e-3
    assert d['gathered_maps_checked'] == 16 and d['gathered_maps_equal_reference_run'] is True, d
    assert d['gpu_equals_reference_run'] is True, d['reference_run']


def test_bench_refuses_a_rank_count_that_is_not_the_launchers():
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'dryrun_bench.py'), '--gpus', '3', '--size', '64', '--steps', '1', '--warmup', '0']
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_PORT='29641')
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and '--gpus 3' in (out.stderr + out.stdout)


def test_bench_eight_self_spawned_ranks_on_a_host_with_as_many_cpus():
    """`bench.py --gpus 8` as the driver's scaling run starts it, on the 8 CPUs of the build container (one per rank: the worker
    threads of a rank are cut to the CPUs it has, the hub serves eight connections, the input-staging probe runs on every rank at
    once): ONE JSON line, whole-job aggregate, every rank's own rate, staging choice and host-memory traffic in it, every gathered
    map checked against the map its sender holds"""
    d = run_bench(8, 4, 29651, size=('--size', '128'), extra=['--probe-seconds', '0.2'])
    assert d['n_gpus'] == 8 and d['steps'] == 4 and d['scaling'] == 'weak' and d['value'] > 0
    assert abs(d['value'] - 8 * 4 * 128 * 128 / (d['ms_per_step'] * 4 / 1e3) / 1e6) / d['value'] < 1e-3
    per_rank = d['per_rank']
    assert len(per_rank['value']) == 8 and all(v > 0 for v in per_rank['value'])
    assert len(per_rank['input_staging']) == 8 and all('probe' in r['chosen_by'] for r in per_rank['input_staging'])
    assert len(per_rank['host_memory_traffic_gb_per_s']) == 8 and all(v > 0 for v in per_rank['host_memory_traffic_gb_per_s'])
    assert d['gathered_maps_checked'] == 8 and d['gathered_maps_equal_senders_own'] is True
    assert 'cpu_baseline' not in d and 'other_configs' not in d or d.get('other_configs') is not None
