"""The ONE stdout line of bench.py is what the driver parses.  Round 4's line had grown to 20 KB, the driver's 8 KB tail cut it
and the round's record came back `parsed: null` (missing 1).  The line is now assembled by bench.compact_line under a
byte budget; these tests build it from the largest record a run can produce -- through the same code -- and parse it."""
import copy
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'ranks_seen', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def full_record():
    """A complete detail record as a default run builds it: every leg present (headline, both sharded legs, the CPU baseline with
    its two extra figures, ten side workloads, c5s and c5u with projections).  Recorded from a real run (profiles/, round 4)."""
    with open(os.path.join(ROOT, 'tests', 'golden', 'bench_detail_sample.json')) as fh:
        return json.load(fh)


def inflate(rec):
    """Worst case: every free-text field ten times as long, three times as many kernels and side workloads, collectives booked."""
    rec = copy.deepcopy(rec)
    rec['dtype'] = rec['dtype'] * 10
    rec['config']['workload'] = rec['config']['workload'] * 10
    rec['cpu_baseline']['sample'] = rec['cpu_baseline']['sample'] * 10
    rec['roofline']['peak_note'] = rec['roofline']['peak_note'] * 10
    for i in range(30):
        rec['kernels']['kernel_with_a_long_name_%02d' % i] = {'ms_per_step': 1.0 / 3.0, 'launches_per_step': 7.0}
        rec['other_workloads']['side_workload_number_%02d' % i] = copy.deepcopy(rec['other_workloads']['c3_D1000000'])
    rec['collectives'] = {'all_reduce_gram_%d' % i: {'calls_per_step': 1.0, 'MB_per_step': 128.0, 'ms_per_step': 1.0 / 7.0}
                          for i in range(12)}
    rec['other_layout'] = copy.deepcopy(rec['other_workloads']['c5s'])
    rec['other_layout']['layout'] = 'clients'
    rec['detail_file'] = 'some/long/path/' * 8 + 'bench_detail.json'
    return rec


def check_line(text, want_all=True):
    assert len(text) < 4096 and '\n' not in text
    line = json.loads(text)
    for key in CONTRACT_KEYS:
        assert key in line, key
    roof, cpu = line['roofline'], line['cpu_baseline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in roof, key
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3 * roof['frac']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in cpu, key
    assert 'workload' in line['config'] and 'model' not in line['config']
    assert abs(line['value'] * line['ms_per_step'] - 1e3) < 1e-3 * 1e3
    return line


def test_default_run_record_fits_with_every_section():
    text = bench.compact_line(full_record())
    line = check_line(text)
    # a realistic record keeps all of its sections: nothing had to be dropped
    for key in ('kernels_ms', 'north_star', 'others', 'projected', 'sharded_path_w1_ms', 'verified_after_timing'):
        assert key in line, key
    star = line['north_star']
    assert star['c5u_target_met'] is False and star['c5s_target_met'] is True     # what round 4 measured
    assert star['c5u']['projected_8'] < 1.0 < star['c5s']['projected_8']
    assert len(text) < 3800          # headroom below the budget


def bench_peak(roof):
    """The peak in the unit of the algorithmic work (bytes/s or flop/s): the record scales it to GB/s or TFLOP/s."""
    return roof['peak'] * (1e9 if roof['unit'] == 'GB/s' else 1e12)


def test_the_line_says_what_each_figure_measured():
    """`others.*.ms` is the wall time of a step while `frac` is computed over the dominant kernel's own time,
    and the clients leg of `sharded_path_w1` runs on a quarter of the columns.  The line now carries the kernel's time beside
    the wall time, the fraction over the wall clock, and every sharded leg's column count."""
    rec = full_record()
    line = json.loads(bench.compact_line(rec))
    for name, leg in line['others'].items():
        src = rec['other_workloads'][name]
        if not src.get('roofline'):
            continue
        assert 'kernel_ms' in leg and 'ms' in leg, name
        roof = src['roofline']
        assert abs(leg['kernel_ms'] - roof['avg_launch_ms'] * roof['launches_per_step']) <= 1e-3 * leg['kernel_ms']
        # `frac` belongs to kernel_ms, not to ms: algorithmic work / kernel time / peak
        want = roof['algorithmic_work_per_launch'] * roof['launches_per_step'] / (leg['kernel_ms'] * 1e-3)
        assert abs(want / (leg['frac'] * bench_peak(roof)) - 1.0) < 5e-3, name
    c3 = line['others']['c3_D1000000']
    assert c3['kernel_ms'] < c3['ms']         # the case VERDICT r5 named: 1.163 ms of kernel inside 1.254 ms of wall
    w1 = line['sharded_path_w1_ms']
    full = rec['config']['params']
    assert w1['columns']['params'] == full and w1['clients']['params'] == full // 4
    scaled = w1['clients']['ms_scaled_to_%d_params' % full]
    assert abs(scaled - 4 * w1['clients']['ms']) <= 1e-3 * scaled
    assert not any(k.startswith('ms_scaled') for k in w1['columns'])
    # the Gram's roofline with the deferred slab update (and the operand split) counted in, next to the tile kernel alone
    roof = line['roofline']
    if 'gram_reduce' in rec['kernels']:
        assert roof['frac_with_slab_update'] < roof['frac']
        assert roof['frac_with_split_and_update'] < roof['frac_with_slab_update']


def test_side_workloads_are_timed_over_a_quarter_second_not_ten_steps():
    """Round 6 (profiles/r06j_short_run_artifact.txt): ten 1 ms rounds after idle measure the chip on its way up, not the round."""
    assert bench.side_leg_steps(1.2e-3, 10) == (209, 20)          # configs[2]'s trimmed mean
    steps, warm = bench.side_leg_steps(30e-6, 10)                   # configs[1]'s Krum: capped
    assert steps == 5000 and warm == 500
    assert bench.side_leg_steps(0.6, 10) == (10, 3)                 # a long round keeps --extras-steps
    assert bench.side_leg_steps(0.0, 0)[0] >= 1


def test_worst_case_record_still_fits_and_keeps_the_contract():
    text = bench.compact_line(inflate(full_record()))
    line = check_line(text)
    assert line['roofline']['kernel'] == 'gram_tile' and line['cpu_baseline']['kind'] == 'port'


def test_minimal_record_no_cpu_baseline_no_extras():
    rec = full_record()
    for key in ('cpu_baseline', 'other_workloads', 'sharded_path_w1', 'projected'):
        rec.pop(key, None)
    line = json.loads(bench.compact_line(rec))
    assert line['cpu_baseline'] is None and 'north_star' not in line and 'others' not in line
    assert line['roofline']['frac'] > 0


def test_multi_gpu_record_with_a_failed_side_leg_still_gives_the_line():
    """An 8-rank record: collectives booked, the other layout's side leg either timed or failed -- the headline's line stands."""
    rec = full_record()
    rec['n_gpus'] = rec['ranks_seen'] = 8
    for key in ('cpu_baseline', 'other_workloads', 'sharded_path_w1', 'projected'):
        rec.pop(key, None)
    rec['collectives'] = {'allreduce_gram': {'calls_per_step': 1.0, 'MB_per_step': 112.0, 'ms_per_step': 1.4},
                          'allgather_output': {'calls_per_step': 1.0, 'MB_per_step': 35.0, 'ms_per_step': 0.3}}
    rec['other_layout'] = {'layout': 'clients', 'error': 'RuntimeError: ' + 'x' * 500}
    line = json.loads(bench.compact_line(rec))
    assert line['n_gpus'] == 8 and line['other_layout']['layout'] == 'clients' and len(line['other_layout']['error']) <= 160
    assert line['collectives_ms']['allreduce_gram'] == 1.4 and line['cpu_baseline'] is None
    rec['other_layout'] = dict(full_record()['other_workloads']['c5s'], layout='clients', steps=3)
    line = json.loads(bench.compact_line(rec))
    assert line['other_layout']['layout'] == 'clients' and line['other_layout']['steps'] == 3 and line['other_layout']['value'] > 0


def test_emit_writes_the_detail_file_and_one_stdout_line(tmp_path, capfd):
    rec = full_record()
    path = str(tmp_path / 'detail.json')
    bench.emit(rec, path)
    out, err = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1 and err == ''
    line = check_line(lines[0])
    assert line['detail_file'] == path
    with open(path) as fh:
        stored = json.load(fh)
    assert 'other_workloads' in stored and 'sharded_path_w1' in stored     # the long sections live here
    # a tree that cannot be written to costs the detail file, never the line
    bench.emit(full_record(), str(tmp_path / 'no_such_dir' / 'detail.json'))
    out, err = capfd.readouterr()
    check_line([ln for ln in out.splitlines() if ln.strip()][-1])
    assert 'detail file not written' in err


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_gpu_bench_prints_one_parsable_line(tmp_path):
    """The bench as the driver runs it (short): the LAST line of stdout parses, fits the budget and carries the contract's keys;
    stdout + stderr together stay inside the driver's 8 KB tail."""
    path = str(tmp_path / 'detail.json')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1',
                           '--clients', '1024', '--params', '262144', '--cpu-seconds', '6', '--no-north-star', '--extras-steps', '2',
                           '--detail-file', path], capture_output=True, text=True, timeout=580)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out_lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    line = check_line(out_lines[-1])
    assert line['n_gpus'] == 1 and line['steps'] == 1 and line['roofline']['kernel'] == 'gram_tile'
    assert line['cpu_baseline']['value'] > 0 and line['others']
    assert len(proc.stdout) + len(proc.stderr) < 8000, (len(proc.stdout), len(proc.stderr))
    assert os.path.isfile(path)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_gpu_bench_rehearses_the_multi_rank_run_on_one_device(tmp_path):
    """`bench.py --gpus 2` end to end on the ONE GPU there is (BYZ_BENCH_ONE_DEVICE=1: both ranks on GPU 0, collectives over
    gloo): the self-spawn under torch.distributed.run, the rank count through the collective library, the barriers and the
    max-over-ranks clock, the column-sharded workload with its Gram all-reduce, rank 0's one line LAST on stdout -- everything
    of the W > 1 path but RCCL itself -- and then the side leg's watchdog: the other layout's point-to-point gathers do not
    complete over gloo, the headline must be out already, the exit status 0 and the detail file marked."""
    path = str(tmp_path / 'detail.json')
    env = dict(os.environ, BYZ_BENCH_ONE_DEVICE='1', BYZ_BENCH_SIDE_LEG_SECONDS='20')
    env.pop('WORLD_SIZE', None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                           '--clients', '1000', '--params', '300000', '--detail-file', path],
                          capture_output=True, text=True, timeout=880, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out_lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(out_lines[-1]) < 4096
    line = json.loads(out_lines[-1])
    for key in CONTRACT_KEYS:
        assert key in line, key
    assert line['cpu_baseline'] is None          # (the CPU baseline is rank 0's at N = 1 only)
    assert abs(line['roofline']['frac'] - line['roofline']['achieved'] / line['roofline']['peak']) < 1e-3 * line['roofline']['frac']
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2 and line['steps'] == 2
    assert 'NOT a multi-GPU measurement' in line['rehearsal']
    assert line['config']['params_per_gpu'] == 150000 and line['scaling'] == 'strong'
    assert line['collectives_ms']['allreduce_gram'] > 0
    assert line['verified_after_timing']['selection_distinct'] == 1000 - 2 * 240
    detail = json.load(open(path))
    leg = detail.get('other_layout') or {}
    # the leg either ran (a record or an error text) or was ended by the watchdog; a hang must have left its marker
    assert leg.get('layout') == 'clients'
    if leg.get('side_leg') == 'timeout':
        assert 'ended by the watchdog' in proc.stderr
