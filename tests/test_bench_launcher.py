"""bench.py's launcher: `python bench.py --gpus N` IS an N-rank run or it fails (it used to run one process and
print n_gpus = N).  The decision is a pure function; the spawn path itself runs here on CPU over gloo with the launcher's
stub workload (no GPU work), two ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan_decisions():
    plan = bench.launch_plan
    assert plan(1, {}, 1) == ('inline', None)
    assert plan(1, {'WORLD_SIZE': '1'}, 1) == ('inline', None)
    # nobody launched ranks: spawn them -- but only when the node has the GPUs
    assert plan(8, {}, 8) == ('spawn', None)
    assert plan(2, {}, 8) == ('spawn', None)
    kind, why = plan(8, {}, 1)
    assert kind == 'fail' and 'exposes 1 GPU' in why
    kind, why = plan(2, {}, 0)
    assert kind == 'fail'
    # launched by torch.distributed.run / the driver: WORLD_SIZE must agree with --gpus, both ways
    assert plan(8, {'WORLD_SIZE': '8'}, 8) == ('inline', None)
    assert plan(8, {'WORLD_SIZE': '1'}, 8)[0] == 'fail'
    assert plan(1, {'WORLD_SIZE': '8'}, 8)[0] == 'fail'
    assert plan(4, {'WORLD_SIZE': '4'}, 2)[0] == 'fail'
    assert plan(0, {}, 8)[0] == 'fail'
    # the CPU stub needs no GPU
    assert plan(2, {}, 0, needs_gpu=False) == ('spawn', None)


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '1'
    return env


def last_json(text):
    lines = [ln for ln in text.strip().splitlines() if ln.startswith('{')]
    assert lines, text
    return json.loads(lines[-1])


@pytest.mark.timeout(300)
def test_plain_invocation_with_two_gpus_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it: two ranks come up (gloo here), the line says what the
    process group says."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'launch-selftest',
                           '--steps', '4', '--warmup', '1'], env=clean_env(), capture_output=True, text=True, timeout=280)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = last_json(proc.stdout)
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['steps'] == 4
    assert line['value'] > 0 and abs(line['ms_per_step'] * line['value'] - 1e3) < 1e-6 * 1e3


@pytest.mark.timeout(120)
def test_one_rank_runs_inline():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--workload', 'launch-selftest',
                           '--steps', '2', '--warmup', '0'], env=clean_env(), capture_output=True, text=True, timeout=100)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = last_json(proc.stdout)
    assert line['n_gpus'] == 1 and line['ranks_seen'] == 1


@pytest.mark.timeout(120)
def test_world_size_that_disagrees_with_gpus_fails_loudly():
    env = clean_env()
    env['WORLD_SIZE'] = '1'
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'launch-selftest'],
                          env=env, capture_output=True, text=True, timeout=100)
    assert proc.returncode != 0
    assert 'WORLD_SIZE' in proc.stderr and not [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_two_gpus_asked_of_a_one_gpu_box_fails_loudly():
    """The exact command form the driver uses, with a GPU count this box does not have: non-zero, no JSON line."""
    torch = pytest.importorskip('torch')
    have = torch.cuda.device_count()
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(have + 1), '--steps', '1', '--warmup', '0'],
                          env=clean_env(), capture_output=True, text=True, timeout=280)
    assert proc.returncode != 0
    assert 'exposes %d GPU' % have in proc.stderr
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith('{')]
