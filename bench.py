#!/usr/bin/env python3
"""Contract bench: `python bench.py --gpus N --steps K --warmup W` -> ONE JSON line on rank 0.

Metric (BASELINE.json): aggregation rounds/sec at N clients x D params on 1/2/4/8 MI355X.

A "step" is one aggregation round of the hot path on device-resident synthetic gradients.  The default
workload is BASELINE.json configs[3] -- Bulyan, N=4000 clients x D=10,000,000 params, f=960 -- the smallest
of the two configurations the metric's "1/2/4/8 GPUs" are quoted on, and one that fits a single 288 GB
MI355X whole (160 GB), so the same total problem is timed at every GPU count (`"scaling": "strong"`).
With `--gpus N` the D columns are sharded N ways (one process per GPU, torch.distributed/RCCL): every rank
runs the MFMA Gram kernel on its column slice, ONE all-reduce of the N x N fp64 Gram is the path's only
exchange step, the selection loop runs replicated, the median-window mean runs on the local columns and the
D-vector is all-gathered (attacking_federate_learning_amd/sharded.py; DESIGN.md section "Multi-GPU").

Other workloads (`--workload`): c2 (Krum N=100, D=79,510 and 21,840), c3 (trimmed_mean N=1000, D=1e6,
trim 200), c5s (attack + Krum + Bulyan, N=10000, one D-slice of 8's worth; the attack's rows are one vector),
c5u (the same with 10,000 distinct rows), attack.  At --gpus 1 all of them ride along in the JSON line under
"other_workloads", c5s / c5u with the projection of BASELINE configs[4] onto eight GPUs.

`--gpus N` IS an N-rank run: started by torch.distributed.run / the driver (WORLD_SIZE == N), or spawned by this
script itself when nobody did; anything else exits non-zero (launch_plan).  `n_gpus` is the process group's size,
`ranks_seen` a sum of ones through the collective library.  At one GPU the line also carries `sharded_path_w1`: the
code path of W > 1 (both layouts) timed with every collective forced through RCCL at world size 1.

Added objects: "roofline" for the dominant kernel (live HIP-event timing around every launch of that kernel on
the stream it is launched on, via the library's byz_timing_* entry points) and "cpu_baseline" (the numpy
oracle -- a port of the reference's defences.py -- timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PEAK_HBM = 8.0e12        # B/s, spec (MI355X_MICROARCH.md chip table; 6.29e12 measured copy)
PEAK_MFMA_F32 = 157.3e12  # flop/s, dense fp32-input MFMA (same table)
PEAK_MFMA_BF16 = 2.5e15   # flop/s, dense bf16 MFMA (same table)
MAL_PROP = 0.24           # reference main.py:106


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--workload', default='c4', choices=['c4', 'c3', 'c2', 'c5s', 'c5u', 'attack', 'launch-selftest'])
    p.add_argument('--clients', type=int, default=None, help='override N')
    p.add_argument('--params', type=int, default=None, help='override total D')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-extras', action='store_true')
    p.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the cpu_baseline leg')
    p.add_argument('--layout', default='columns', choices=['columns', 'clients', 'both'],
                   help='multi-GPU layout of the gradient matrix (sharded.py); "both" times the other one as well')
    p.add_argument('--no-sharded-w1', action='store_true',
                   help='skip the leg that times the W > 1 code path (collectives forced) on this one GPU')
    p.add_argument('--no-north-star', action='store_true', help='skip the c5s / c5u legs of the default run')
    p.add_argument('--detail-file', default='bench_detail.json',
                   help='where the full record goes (every leg, every kernel table); stdout carries only the compact line')
    p.add_argument('--extras-steps', type=int, default=10, help='timed rounds of each small side workload')
    p.add_argument('--north-star-steps', type=int, default=2, help='timed rounds of the c5s / c5u legs')
    return p.parse_args(argv)


# ---- launching: `python bench.py --gpus N` must BE an N-rank run or fail -----------------------------------------
def launch_plan(gpus, env, device_count, needs_gpu=True):
    """What `python bench.py --gpus N` does, decided from the environment alone (a pure function: tests/test_bench_launcher.py).

    ('inline', None)   this process is one rank: either started by torch.distributed.run / the driver (WORLD_SIZE set and
                       equal to N), or N == 1
    ('spawn', None)    N > 1 and nobody launched ranks: re-exec under torch.distributed.run with N ranks on this node
    ('fail', message)  WORLD_SIZE disagrees with --gpus, or the box exposes fewer than N GPUs: a line that says n_gpus = N
                       must never come from fewer than N ranks (`--gpus 8` without torchrun ran ONE process
                       and printed n_gpus 8)"""
    if gpus < 1:
        return 'fail', '--gpus must be >= 1'
    world_env = env.get('WORLD_SIZE')
    if world_env is not None:
        if int(world_env) != gpus:
            return 'fail', '--gpus %d but WORLD_SIZE=%s: the launcher started another number of ranks' % (gpus, world_env)
        if needs_gpu and device_count < gpus:
            return 'fail', '--gpus %d but this node exposes %d GPU(s)' % (gpus, device_count)
        return 'inline', None
    if gpus == 1:
        return 'inline', None
    if needs_gpu and device_count < gpus:
        return 'fail', '--gpus %d but this node exposes %d GPU(s): refusing to print an n_gpus=%d line from fewer ranks' % (
            gpus, device_count, gpus)
    return 'spawn', None


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def spawn_ranks(gpus, argv):
    """Re-exec this script as `gpus` ranks of one node (one process per GPU, rendezvous on 127.0.0.1) and pass the child's
    output and exit code through."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def ranks_seen(torch, dist, device):
    """How many ranks really take part: every rank contributes a one through the collective library itself."""
    if not dist.is_initialized():
        return 1
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())


def launch_selftest(args):
    """The launcher's own workload (CPU, gloo): K trivial steps timed as the contract says, so that the spawn path, the
    rank count and the max-over-ranks clock are exercised where there is no GPU (tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    seen = ranks_seen(torch, dist, torch.device('cpu'))
    acc = torch.zeros(1 << 16, dtype=torch.float64)
    for _ in range(args.warmup):
        acc += 1.0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc += float(rank + 1)
        if world > 1:
            dist.all_reduce(acc)     # the stub's exchange step
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({'metric': 'launcher self-test steps/sec', 'value': args.steps / elapsed, 'unit': 'steps/s',
                          'n_gpus': dist.get_world_size() if dist.is_initialized() else 1, 'ranks_seen': seen,
                          'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
                          'data': 'synthetic', 'config': {'workload': 'launch-selftest (CPU, gloo): no GPU work'}}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


# ---- synthetic inputs -----------------------------------------------------------------------------
def make_matrix(torch, n, d, seed, device):
    """'scaled' family of SURVEY.md 8(d): G[i,:] = s_i * N(0,1), s_i = 1 + 0.5*pi(i)/N: well-separated Krum
    scores.  Generated on the device in row blocks (no second copy of a 160 GB matrix)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    g = torch.empty((n, d), dtype=torch.float32, device=device)
    perm = torch.from_numpy(np.random.default_rng(seed).permutation(n)).to(device)
    scale = (1.0 + 0.5 * perm.to(torch.float32) / n)
    rows_per = max(1, (1 << 28) // max(d, 1))
    for r in range(0, n, rows_per):
        blk = g[r:r + rows_per]
        blk.normal_(generator=gen)
        blk.mul_(scale[r:r + rows_per, None])
    return g


def column_bounds(d_total, world):
    base, extra = divmod(d_total, world)
    out, start = [], 0
    for r in range(world):
        stop = start + base + (1 if r < extra else 0)
        out.append((start, stop))
        start = stop
    return out


# ---- workloads ------------------------------------------------------------------------------------
class Workload:
    name = ''
    defence = ''

    def dtype(self):
        return 'f32'

    def at_profiled_size(self):
        return True

    def describe(self):
        raise NotImplementedError


class BulyanSharded(Workload):
    """configs[3] (and the Bulyan half of configs[4]): column-sharded Bulyan through ShardedAggregator."""

    def __init__(self, torch, agg, eng, n, d_total, device, seed, with_attack=False, layout='columns', distinct=False, g=None):
        self.torch, self.agg, self.eng = torch, agg, eng
        self.n, self.d_total = n, d_total
        self.f = int(n * MAL_PROP)
        self.layout = layout
        if layout == 'clients':
            # north_star's layout: rank r holds the rows of its clients, all D columns (reference main.py:26-32)
            self.rows_per_rank = [n // agg.world + (1 if r < n % agg.world else 0) for r in range(agg.world)]
            self.d_local = d_total
            shape = (self.rows_per_rank[agg.rank], d_total)
        else:
            lo, hi = column_bounds(d_total, agg.world)[agg.rank]
            self.d_local = hi - lo
            shape = (n, self.d_local)
        if g is not None:     # a matrix the caller already holds (at one rank both layouts are the whole matrix)
            assert tuple(g.shape) == shape
            self.g = g
        else:
            self.g = make_matrix(torch, shape[0], shape[1], seed + 17 * agg.rank, device)
        self.with_attack = with_attack
        # distinct: the attack's column statistics are computed but the malicious rows are NOT overwritten with the one
        # drifted vector -- all N rows stay distinct (an attacker who adds per-row noise; nothing for the identical-row
        # shortcut to fold): configs[4] without the shortcut
        self.distinct = bool(distinct and with_attack)
        self.name = ('c5u' if self.distinct else 'c5s') if with_attack else 'c4'
        self.defence = 'attack+Krum+Bulyan' if with_attack else 'Bulyan'
        self.last = None

    def step(self):
        if self.layout == 'clients':
            if self.with_attack:
                self.agg.drift_attack_clients(self.g, self.rows_per_rank, self.f, 1.5, write_back=not self.distinct)
                dist_m = self.agg.client_distances(self.g, self.rows_per_rank)
                idx, sel = self.agg.kernels.krum_bulyan_select(dist_m, self.n, self.f)    # one sort of the rows for both
                sel = np.asarray(sel, dtype=np.int64)
                cols, row_index = self.agg.reshard_rows_to_columns(self.g, self.rows_per_rank, sel)
                out = self.agg.kernels.trimmed_mean(cols, 2 * self.f, row_index=row_index)
                self.last = (self.agg._maybe_gather(out, True, total=self.d_total), sel, idx)
            else:
                self.last = self.agg.bulyan_clients(self.g, self.rows_per_rank, self.n, self.f, return_selection=True)
            return
        if self.with_attack:
            # rows 0..m-1 are the malicious clients (reference main.py:28); per column, no exchange
            self.agg.drift_attack(self.g, self.f, 1.5, write_back=not self.distinct)
            dist_m = self.agg.global_distances(self.g)
            # Krum's index and Bulyan's selection from the one distance matrix, one sort of its rows; the selection stays on the device
            idx, sel = self.agg.kernels.krum_bulyan_select(dist_m, self.n, self.f, on_device=True)
            out = self.agg.kernels.trimmed_mean(self.g, 2 * self.f, row_index=sel)
            self.last = (self.agg._maybe_gather(out, True, total=self.d_total), sel, idx)
        else:
            out, sel = self.agg.bulyan(self.g, self.n, self.f, gather=True, return_selection=True,
                                       total_columns=self.d_total)
            self.last = (out, sel)

    def verify(self):
        """After the timed region: the last step's selection holds theta distinct clients and its output is finite (the
        full-size parity checks proper are tests/test_gpu_fullsize.py; this only refuses to print a rate for garbage)."""
        out, sel = self.last[0], self.last[1]
        sel = np.asarray(sel.numpy() if hasattr(sel, 'numpy') else sel).reshape(-1)
        theta = self.n - 2 * self.f
        if len(sel) != theta or len(set(sel.tolist())) != theta or sel.min() < 0 or sel.max() >= self.n:
            raise SystemExit('bench: the selection of the last step is not %d distinct clients' % theta)
        if not bool(self.torch.isfinite(out).all()):
            raise SystemExit('bench: the aggregated vector of the last step is not finite')
        return {'selection_distinct': theta, 'output_finite': True}

    def dtype(self):
        # fp32 data and fp32 results; for N > 256 the Gram contraction runs as an exact three-way bf16 split of every
        # fp32 operand on the bf16 matrix cores (six bf16 MFMAs per fp32 product block, fp32 accumulate)
        if self.n <= 256:
            return 'f32'
        mode = self.dominant().get('arithmetic')
        return {'f16x2': 'f32 (Gram: fp16x2 split of every fp32 operand, 3 fp16 MFMAs per block, fp32/fp64 accumulate; <= 2e-7 of |gi||gj| vs fp64, distances <= 1e-6)',
                'split': 'f32 (Gram: bf16x3 exact-split MFMA, fp32 accumulate)'}.get(mode, 'f32')

    def dominant(self):
        # the Gram: N^2 * D_local flops per launch (half Gram, 2 flop per MAC) -- SURVEY.md 8(d).  The peak is the
        # dense fp32-input MFMA peak: the algorithmic flops are fp32 flops, whatever instructions carry them.
        # under the attack rows 0..f-1 are one vector and the engine runs the Gram over the unique rows only: the
        # kernel's work is (N - f + 1)^2 * D_local, not N^2 * D_local
        rows = self.n - self.f + 1 if self.with_attack and not self.distinct and self.n >= 512 else self.n
        share = self.agg.world if self.layout == 'clients' else 1     # clients: every rank does 1/W of the tiles, all D
        # Which arithmetic the engine picks (csrc/gram.hip launch_gram_rows): few tiles -> fp32-input MFMA; many tiles ->
        # bf16 x 3 (six bf16 MFMA flops per fp32 flop); many tiles AND a long K -> operands split once into two fp16 planes
        # (csrc/gram_planes.hip), three fp16 MFMA flops per fp32 flop.  The roof of each arithmetic is the dense 16-bit
        # MFMA peak divided by that factor; `mfma_issued` says how busy the matrix pipe really is.
        t = -(-rows // 128)
        n_tiles = -(-(t * (t + 1) // 2) // share)
        mode = os.environ.get('BYZ_GRAM_MODE')
        long_k = n_tiles >= 256 and self.d_local > 16384 and os.environ.get('BYZ_GRAM_PLANES', '1') != '0'
        if mode is None:
            mode = 'f16x2' if long_k else ('split' if n_tiles >= 4 else 'exact')
        if mode == 'f16x2' and not long_k:
            mode = 'split'
        factor = {'exact': 1.0, 'split': 6.0, 'f16x2': 3.0}[mode]
        peak = PEAK_MFMA_F32 if mode == 'exact' else PEAK_MFMA_BF16 / factor
        note = {'exact': 'dense fp32-input MFMA peak',
                'split': 'fp32-equivalent roof of the exact bf16x3 split: 2.5 PF dense bf16 / 6 MFMA flops per fp32 flop',
                'f16x2': 'fp32-equivalent roof of the fp16x2 split (operands split once, csrc/gram_planes.hip): '
                         '2.5 PF dense fp16 / 3 MFMA flops per fp32 flop'}[mode]
        return {'kernel': 'gram_tile', 'bound': 'mfma', 'work': float(rows) ** 2 * self.d_local / share,
                'peak': peak, 'unit': 'TFLOP/s', 'scale': 1e12, 'arithmetic': mode, 'work_is_per_step': True,
                'issued_factor': factor, 'issued_peak': PEAK_MFMA_BF16 if mode != 'exact' else PEAK_MFMA_F32,
                'companion': 'plane_split' if long_k and mode in ('f16x2', 'split') else None,
                'peak_note': note}

    def at_profiled_size(self):
        return self.name == 'c4' and self.n == 4000 and self.d_local == 10000000

    def config(self):
        rows = ('' if not self.with_attack else
                ', all %d rows distinct (attack statistics computed, rows not overwritten)' % self.n if self.distinct else
                ', the %d malicious rows one vector (malicious.py:26-27): Gram over %d unique rows' % (self.f, self.n - self.f + 1))
        return {'workload': '%s: %s N=%d D=%d f=%d theta=%d (BASELINE configs[%d]%s), %s sharded %d-way%s'
                            % (self.name, self.defence, self.n, self.d_total, self.f, self.n - 2 * self.f,
                               4 if self.with_attack else 3,
                               ": one GPU's slice of eight, D = 25M / 8" if self.with_attack and self.d_total == 3_125_000 * self.agg.world else '',
                               self.layout, self.agg.world, rows),
                'clients': self.n, 'params': self.d_total, 'corrupted': self.f, 'layout': self.layout,
                'params_per_gpu': self.d_local, 'input_family': 'scaled', 'distinct_rows': self.n if (self.distinct or not self.with_attack) else self.n - self.f + 1}


class TrimmedMeanC3(Workload):
    name, defence = 'c3', 'TrimmedMean'

    def __init__(self, torch, eng, n, d, device, seed):
        self.eng, self.n, self.d, self.c = eng, n, d, n // 5
        self.g = make_matrix(torch, n, d, seed, device)

    def step(self):
        self.last = self.eng.trimmed_mean(self.g, self.n, self.c)

    def dominant(self):
        return {'kernel': 'trimmed_mean', 'bound': 'hbm', 'work': 4.0 * self.n * self.d + 4.0 * self.d,
                'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9}

    def at_profiled_size(self):
        return self.n == 1000 and self.d == 1000000

    def config(self):
        return {'workload': 'c3: TrimmedMean N=%d D=%d trim=%d (BASELINE configs[2])' % (self.n, self.d, self.c),
                'clients': self.n, 'params': self.d, 'corrupted': self.c}


class KrumC2(Workload):
    name, defence = 'c2', 'Krum'

    def __init__(self, torch, eng, n, d, device, seed):
        self.eng, self.n, self.d, self.f = eng, n, d, int(n * MAL_PROP)
        self.g = make_matrix(torch, n, d, seed, device)

    def step(self):
        # asynchronous (the winning row is copied on the device); timed_steps() ends with eng.check(), which raises if any
        # kernel of the timed rounds flagged a failure
        self.last = self.eng.krum(self.g, self.n, self.f, check=False)

    def dominant(self):
        # N <= 128 runs csrc/krum_small.hip (K-sliced fp16x2 Gram, five launches) unless BYZ_KRUM_SMALL=0; its first kernel
        # reports under the same timing slot as the general path's Gram tiles
        small = (self.n <= 128 and self.d <= int(os.environ.get('BYZ_KRUM_SMALL_MAX_COLS', 262144))
                 and os.environ.get('BYZ_KRUM_SMALL', '1') != '0')
        return {'kernel': 'gram_tile', 'bound': 'hbm', 'work': 4.0 * self.n * self.d + 4.0 * self.n * self.n,
                'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9, 'arithmetic': 'f16x2' if small else 'exact',
                # a matrix of N D 4 < 256 MiB lives in the Infinity Cache across rounds: the fraction below is against the HBM
                # peak because the contract has no other name for it, but what bounds these rounds is cache latency and
                # the four dependent kernel boundaries behind the Gram
                'bound_note': 'cache / latency (matrix resident in the 256 MiB Infinity Cache)' if 4.0 * self.n * self.d < (256 << 20) else None,
                'peak_note': ('csrc/krum_small.hip: K-sliced fp16x2 Gram over all rows (HBM bound: N/4 flop per byte)' if small
                              else 'general path: fp32-input MFMA Gram tiles')}

    def at_profiled_size(self):
        return self.n == 100 and self.d == 79510

    def config(self):
        return {'workload': 'c2: Krum N=%d D=%d f=%d (BASELINE configs[1])' % (self.n, self.d, self.f),
                'clients': self.n, 'params': self.d, 'corrupted': self.f}


class AttackOnly(Workload):
    name, defence = 'attack', 'DriftAttack'

    def __init__(self, torch, eng, m, d, device, seed):
        self.eng, self.m, self.d = eng, m, d
        self.g = make_matrix(torch, m, d, seed, device)

    def step(self):
        self.last = self.eng.drift_attack(self.g, 1.5, write_back=False)

    def dominant(self):
        return {'kernel': 'column_stats', 'bound': 'hbm', 'work': 4.0 * self.m * self.d + 4.0 * self.d,
                'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9}

    def at_profiled_size(self):
        return self.m == 2400 and self.d == 1000000

    def config(self):
        return {'workload': 'attack: A Little Is Enough over m=%d malicious rows, D=%d' % (self.m, self.d),
                'clients': self.m, 'params': self.d}


class NoDefense(Workload):
    """the column mean (defences.py:13-14) -- the same streaming kernel as the attack's statistics."""
    name, defence = 'no_defense', 'NoDefense'

    def __init__(self, torch, eng, n, d, device, seed):
        self.eng, self.n, self.d = eng, n, d
        self.g = make_matrix(torch, n, d, seed, device)

    def step(self):
        self.last = self.eng.no_defense(self.g, self.n, 0)

    def dominant(self):
        return {'kernel': 'column_stats', 'bound': 'hbm', 'work': 4.0 * self.n * self.d + 4.0 * self.d,
                'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9}

    def config(self):
        return {'workload': 'no_defense: column mean over N=%d clients, D=%d (defences.py:13-14)' % (self.n, self.d),
                'clients': self.n, 'params': self.d}


class ServerUpdate(Workload):
    """velocity = momentum * velocity - lr * agg; weights += velocity (server.py:89-90), one fused
    launch on device-resident vectors: 12 bytes read + 8 written per parameter."""
    name, defence = 'server_update', 'momentum step'

    def __init__(self, torch, eng, d, device, seed):
        self.eng, self.d = eng, d
        gen = torch.Generator(device=device).manual_seed(seed)
        self.w = torch.randn(d, device=device, generator=gen)
        self.v = torch.zeros(d, device=device)
        self.agg = torch.randn(d, device=device, generator=gen)

    def step(self):
        self.eng.server_update(self.w, self.v, self.agg, 0.9, 0.1)
        self.last = self.w

    def dominant(self):
        return {'kernel': 'misc', 'bound': 'hbm', 'work': 20.0 * self.d, 'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9}

    def config(self):
        return {'workload': 'server update: fused momentum step on D=%d device-resident parameters (server.py:89-90)' % self.d,
                'params': self.d}


class BackdoorHook(Workload):
    """the two vector steps of BackdoorAttack._attack_grads (backdoor.py:54, 57-63)."""
    name, defence = 'backdoor', 'BackdoorAttack'

    def __init__(self, torch, eng, d, device, seed):
        self.eng, self.d = eng, d
        gen = torch.Generator(device=device).manual_seed(seed)
        self.mean, self.params, self.mal = (torch.randn(d, device=device, generator=gen) for _ in range(3))
        self.std = torch.rand(d, device=device, generator=gen)

    def step(self):
        start = self.eng.backdoor_initial_params(self.params, self.mean, 0.1)
        self.last = self.eng.backdoor_clip(self.mean, self.std, self.params, self.mal, 0.1, 1.5), start

    def dominant(self):   # two launches per step: 12 + 20 bytes per element, 16 on average
        return {'kernel': 'misc', 'bound': 'hbm', 'work': 16.0 * self.d, 'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9}

    def at_profiled_size(self):
        return False

    def config(self):
        return {'workload': 'backdoor hook: start parameters + clipped gradient, D=%d' % self.d, 'params': self.d}


class Assembly(Workload):
    """N clients' per-parameter device gradients -> rows of the device-resident matrix."""
    name, defence = 'assembly', 'collect_gradients'

    def __init__(self, torch, eng, n, shapes, device, seed, batched=False):
        from attacking_federate_learning_amd.assembly import GradientMatrix
        self.eng, self.n = eng, n
        self.d = sum(int(np.prod(sh)) for sh in shapes)
        gen = torch.Generator(device=device).manual_seed(seed)
        # every client's own .grad tensors, as N models that stepped on the GPU leave them
        self.users = [type('Client', (), {'grads': [torch.randn(sh, device=device, generator=gen) for sh in shapes]})()
                      for _ in range(n if not batched else 0)]
        self.n_tensors = len(shapes)
        self.matrix = GradientMatrix(n, self.d, engine=eng, torch_device=device)
        self.batched = None
        if batched:   # what a batched client step hands over: (n, *shape) per parameter
            self.batched = [torch.randn((n,) + tuple(sh), device=device, generator=gen) for sh in shapes]

    def step(self):
        if self.batched is not None:
            self.matrix.set_all(self.batched)
            return
        self.matrix.collect_gradients(self.users)      # server.py:81-83; one launch for all device-resident clients

    def dominant(self):   # per launch: the rows it fills are read once and written once
        rows = self.n
        return {'kernel': 'misc', 'bound': 'hbm', 'work': 8.0 * self.d * rows, 'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9}

    def at_profiled_size(self):
        return False

    def config(self):
        return {'workload': 'gradient assembly (%s): %d clients x %d tensors -> device matrix, D=%d' % (
                    'one launch, batched gradients' if self.batched is not None else
                    "collect_gradients over every client's own tensors: one launch, pointer table on the device", self.n, self.n_tensors, self.d),
                'clients': self.n, 'params': self.d}


class ClientStep(Workload):
    """every client's forward/backward from the same weights in one batched pass
    (torch.func.vmap; rocBLAS batched GEMMs, not a libbyzagg kernel) + assembly into the device matrix."""
    name, defence = 'client_step', 'dispatch_weights+collect_gradients'

    def __init__(self, torch, eng, n, batch, device, seed):
        from attacking_federate_learning_amd.assembly import GradientMatrix
        self.eng, self.n, self.batch, self.d = eng, n, batch, 79510

        class Net(torch.nn.Module):   # the shape of the reference's MnistNet (data_sets.py:13-23)
            def __init__(self):
                super().__init__()
                self.fc1, self.fc2 = torch.nn.Linear(784, 100), torch.nn.Linear(100, 10)

            def forward(self, x):
                return torch.log_softmax(self.fc2(torch.relu(self.fc1(x))), dim=1)

        torch.manual_seed(seed)
        self.net = Net().to(device)
        gen = torch.Generator(device=device).manual_seed(seed)
        self.weights = 0.05 * torch.randn(self.d, device=device, generator=gen)
        self.data = torch.randn((n, batch, 784), device=device, generator=gen)
        self.target = torch.randint(0, 10, (n, batch), device=device, generator=gen)
        self.matrix = GradientMatrix(n, self.d, engine=eng, torch_device=device)

    def step(self):
        from attacking_federate_learning_amd.clients import collect_batched
        collect_batched(self.matrix, self.net, self.weights, self.data, self.target)

    def dominant(self):   # the libbyzagg part of the step is the one assembly launch
        return {'kernel': 'misc', 'bound': 'hbm', 'work': 8.0 * self.d * self.n, 'peak': PEAK_HBM, 'unit': 'GB/s', 'scale': 1e9}

    def at_profiled_size(self):
        return False

    def config(self):
        return {'workload': 'batched client step: %d MnistNet clients x batch %d, gradients into the device matrix' % (self.n, self.batch),
                'clients': self.n, 'params': self.d}


# ---- timing ---------------------------------------------------------------------------------------
def timed_steps(torch, dist, wl, eng, steps, warmup, world, events=True, agg=None):
    for _ in range(warmup):
        wl.step()
    if agg is not None:
        torch.cuda.synchronize()
        agg.comm_report()     # the collectives' books cover the timed steps only
    eng.timing(events)    # HIP events around every kernel launch, on the launch stream
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    eng.check()           # the sticky device status word: a kernel of the timed steps that flagged a failure raises here
    per_kernel = eng.timing_read()
    eng.timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, per_kernel


def roofline_of(wl, per_kernel, traffic_table, steps=None):
    """`work` is the dominant kernel's algorithmic work per STEP; a step may take several launches of it (the long-K Gram
    runs one launch per super-chunk of columns), so achieved = work per step / that kernel's time per step, and the
    per-launch figures are the per-step ones divided by the launches per step."""
    dom = wl.dominant()
    k = per_kernel.get(dom['kernel'])
    if not k:
        return None
    if not steps or not dom.get('work_is_per_step'):
        steps = k['launches']          # `work` is per launch (every other workload: one dominant launch per unit of work)
    per_step = k['launches'] / float(steps)
    step_s = k['total_ms'] / steps / 1e3
    achieved = dom['work'] / step_s
    traffic = None
    if traffic_table:
        # the committed PMC passes were taken at the BASELINE sizes of each workload; other sizes report null
        rec = traffic_table.get('%s/%s' % (wl.name, dom['kernel'])) if wl.at_profiled_size() else None
        if rec and rec.get('arithmetic', 'split') == dom.get('arithmetic', 'split') and traffic_record_is_current(rec):
            traffic = rec.get('hbm_bytes_per_launch')
    out = {'kernel': dom['kernel'], 'bound': dom['bound'], 'achieved': achieved / dom['scale'],
           'peak': dom['peak'] / dom['scale'], 'unit': dom['unit'], 'frac': achieved / dom['peak'],
           'traffic': traffic, 'avg_launch_ms': step_s * 1e3 / per_step, 'launches': k['launches'],
           'launches_per_step': per_step, 'algorithmic_work_per_launch': dom['work'] / per_step}
    if dom.get('arithmetic'):
        out['arithmetic'] = dom['arithmetic']
    if dom.get('peak_note'):
        out['peak_note'] = dom['peak_note']
    if dom.get('bound_note'):
        out['bound_note'] = dom['bound_note']
    comp = per_kernel.get(dom.get('companion') or '')
    # the deferred slab update (chunk_reduce_kernel) was INSIDE the tile kernel until round 5: its time belongs to the Gram when
    # rounds are compared (leaving it out flattered the round-over-round figure by ~2 %)
    red = per_kernel.get('gram_reduce') if dom['kernel'] == 'gram_tile' else None
    red_ms = red['total_ms'] if red and red['launches'] else 0.0
    if red_ms:
        with_s = (k['total_ms'] + red_ms) / steps / 1e3
        out['with_slab_update'] = {'achieved': dom['work'] / with_s / dom['scale'], 'unit': dom['unit'],
                                   'frac': dom['work'] / with_s / dom['peak'], 'gram_reduce_ms_per_step': red_ms / steps}
    if comp and comp['launches']:
        # the one-off operand split that feeds the tile kernel (HBM bound): the rate with its time (and the slab update's)
        # counted in
        both_s = (k['total_ms'] + comp['total_ms'] + red_ms) / steps / 1e3
        out['with_plane_split'] = {'achieved': dom['work'] / both_s / dom['scale'], 'unit': dom['unit'],
                                   'frac': dom['work'] / both_s / dom['peak'],
                                   'plane_split_ms_per_step': comp['total_ms'] / steps}
    if dom.get('issued_factor', 1.0) != 1.0:
        out['vs_fp32_mfma_peak'] = achieved / PEAK_MFMA_F32
        # the matrix-core instructions actually issued per algorithmic fp32 flop, against the dense 16-bit MFMA peak
        out['mfma_issued'] = {'achieved': achieved * dom['issued_factor'] / 1e12, 'peak': dom['issued_peak'] / 1e12,
                              'unit': 'TFLOP/s (16-bit MFMA)', 'frac': achieved * dom['issued_factor'] / dom['issued_peak']}
    return out


def side_leg_steps(seconds_per_step, minimum, target_seconds=0.25, most=5000):
    """(timed steps, warm-up steps) of a short side workload: at least `minimum` steps and about `target_seconds` of them."""
    per = max(float(seconds_per_step), 1e-7)
    steps = int(min(max(int(minimum), 1, math.ceil(target_seconds / per)), most))
    return steps, max(3, steps // 10)


def kernel_table(per_kernel, steps):
    return {name: {'ms_per_step': round(v['total_ms'] / steps, 4), 'launches_per_step': v['launches'] / steps}
            for name, v in per_kernel.items()}


XGMI_LINK = 153e9          # B/s per direction and link (MI355X_MICROARCH.md; 7 links per GPU, point to point)


def projected_scaling(wl, kernels, ms_per_step):
    """PROJECTION, not a measurement (one GPU is all this bench can see unless the driver starts it on more):
    the columns layout's round on W GPUs from this GPU's measured per-kernel times.  Per column and therefore divided by W:
    the Gram tiles, the operand split, the second-stage trimmed mean, the column statistics of the attack.  Replicated on
    every rank, not divided: the row sorts and the Bulyan selection loop.  Added: ONE ring all-reduce of the N x N fp64 Gram
    (2 (W - 1) / W of its bytes through one xGMI link: the per-link bound, no credit for the mesh's other six links) and the
    all-gather of the D-vector.  For configs[4]'s slice (c5s) the measured shard already IS one of eight: W = 8 only."""
    per_column = sum(kernels.get(k, {}).get('ms_per_step', 0.0) for k in ('gram_tile', 'plane_split', 'trimmed_mean', 'column_stats',
                                                                           'gram_reduce'))
    kernel_total = sum(v['ms_per_step'] for v in kernels.values())
    outside = max(ms_per_step - kernel_total, 0.0)        # host gaps, memsets, launches between the kernels
    replicated = kernel_total - per_column
    gram_bytes = 8.0 * wl.n * wl.n
    out = {'label': 'PROJECTED from this GPU\'s measured kernels + an xGMI link model; not measured on W GPUs',
           'layout': 'columns', 'link_GBps': XGMI_LINK / 1e9, 'per_column_ms': per_column, 'replicated_ms': replicated + outside}
    shard_is_one_of = 8 if wl.with_attack else 1
    for w in ((8,) if wl.with_attack else (2, 4, 8)):
        split = w // shard_is_one_of
        allreduce = 2.0 * (w - 1) / w * gram_bytes / XGMI_LINK * 1e3
        allgather = (w - 1) / w * 4.0 * wl.d_total * shard_is_one_of / XGMI_LINK * 1e3
        ms = per_column / split + replicated + outside + allreduce + allgather
        out['gpus_%d' % w] = {'ms_per_step': ms, 'rounds_per_s': 1e3 / ms, 'allreduce_gram_ms': allreduce, 'allgather_output_ms': allgather,
                              'params_total': wl.d_total * shard_is_one_of}
    return out


def traffic_record_is_current(rec):
    """A PMC record stands for the kernel it was taken on: it names the kernel's source file and that file's hash at the
    time (scripts/update_traffic.py).  A kernel edited since reports `traffic: null` rather than a stale number; records
    from before round 3 carry no hash and are taken as they are (their kernels have not changed)."""
    src, want = rec.get('source'), rec.get('source_sha16')
    if not src or not want:
        return True
    try:
        import hashlib
        path = os.path.join(ROOT, 'attacking_federate_learning_amd', 'csrc', src)
        return hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16] == want
    except OSError:
        return False


def load_traffic_table():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*traffic*.json), doubled on the
    read side as MI355X_MICROARCH.md section HBM prescribes for gfx950.  None when no pass has been recorded."""
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


# ---- CPU baseline (oracle = restatement of the reference; checker/baseline only, never the product) ---------------
def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(wl, budget_s):
    """The reference's CPU path on this box's host cores, on a bounded sample of the same workload.

    Three figures, all labelled:
      value / kind "port"   oracle.faithful (vectorised numpy restatement), OpenBLAS pinned to ONE thread
      also.port_all_cores   the same with OpenBLAS free to use the box (the reference's only threaded call is sdot)
      also.as_shipped       oracle.shipped: the reference's own loops (dicts, sorted(), sum, key=abs lambda), one thread
    """
    from oracle import faithful, shipped
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    rng = np.random.default_rng(5)
    t_budget = max(budget_s, 6.0)
    cores = usable_cores()

    def timed(fn, seconds, at_least=1):
        t0, reps = time.perf_counter(), 0
        while reps < at_least or time.perf_counter() - t0 < seconds:
            fn()
            reps += 1
        return (time.perf_counter() - t0) / reps, reps

    def bulyan_parts(threads, as_shipped, seconds):
        """(round time, description) of configs[3]/[4] from its three measured parts."""
        n, d, f = wl.n, wl.d_total, wl.f
        theta = n - 2 * f
        rows = [rng.standard_normal(d).astype(np.float32) for _ in range(3)]
        # (i) one distance at the true D (defences.py:20): norm of the fp32 difference
        t_pair, pairs = timed(lambda: np.linalg.norm(rows[0] - rows[1]), 0.3 * seconds)
        # (ii) one Krum pick over n live rows (defences.py:32-37)
        pts = rng.standard_normal((n, 8)).astype(np.float32)
        dist = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)).astype(np.float32)
        np.fill_diagonal(dist, np.inf)
        if as_shipped:
            as_dict = shipped.dict_from_dense(dist)
            t0 = time.perf_counter()
            shipped.krum(None, n, f, distances=as_dict, return_index=True)
        else:
            t0 = time.perf_counter()
            faithful.krum_pick(dist, faithful.visit_order(n), n, f)
        t_pick = time.perf_counter() - t0
        # (iii) trimmed_mean per column over theta rows (defences.py:48-51)
        cols = 48 if as_shipped else 400
        sub = rng.standard_normal((theta, cols)).astype(np.float32)
        t0 = time.perf_counter()
        (shipped if as_shipped else faithful).trimmed_mean(sub, theta, 2 * f)
        t_col = (time.perf_counter() - t0) / cols
        # picks shrink: pick t scores n - t rows of n - t - 1 distances
        pick_total = sum(t_pick * ((n - t) / n) ** 2 for t in range(theta))
        total = t_pair * n * (n - 1) / 2 + pick_total + t_col * d + (t_pick if wl.with_attack else 0.0)
        return total, ('extrapolated: %.1f ms per distance at D=%d x N(N-1)/2 (%d timed); one Krum pick at N=%d %.2f s x '
                       'theta picks with the (n_t/N)^2 shrink; trimmed_mean %.0f us per column at theta=%d x D (%d timed)'
                       % (t_pair * 1e3, d, pairs, n, t_pick, t_col * 1e6, theta, cols))

    def measure(threads, as_shipped, seconds):
        def run():
            if isinstance(wl, BulyanSharded):
                total, sample = bulyan_parts(threads, as_shipped, seconds)
                return 1.0 / total, sample
            if isinstance(wl, TrimmedMeanC3):
                cols = 24 if as_shipped else 200
                sub = rng.standard_normal((wl.n, cols)).astype(np.float32)
                fn = (shipped if as_shipped else faithful).trimmed_mean
                t, reps = timed(lambda: fn(sub, wl.n, wl.c), seconds)
                return 1.0 / (t / cols * wl.d), 'extrapolated from %d columns at N=%d (%.0f us/column x D)' % (
                    cols * reps, wl.n, t / cols * 1e6)
            if isinstance(wl, KrumC2):
                g = rng.standard_normal((wl.n, wl.d)).astype(np.float32)
                fn = (shipped if as_shipped else faithful).krum
                t, reps = timed(lambda: fn(g, wl.n, wl.f), seconds)
                return 1.0 / t, 'full rounds: %d timed' % reps
            if isinstance(wl, AttackOnly):
                m = min(wl.m, 64)
                g = rng.standard_normal((m, min(wl.d, 1 << 20))).astype(np.float32)
                t, reps = timed(lambda: faithful.drift_vector(g, 1.5), 0.5 * seconds)
                return 1.0 / (t * (wl.m / m) * (wl.d / g.shape[1])), 'extrapolated linearly from m=%d, D=%d' % (m, g.shape[1])
            return None, 'n/a'
        if threadpool_limits is not None:
            with threadpool_limits(limits=threads):
                return run()
        return run()

    value, sample = measure(1, False, 0.45 * t_budget)
    out = {'value': value, 'unit': 'rounds/s', 'cores': 1, 'kind': 'port', 'sample': sample,
           'host_cores_available': cores, 'also': {}}
    many = min(cores, 64)     # scipy-openblas is built for up to 64 threads
    v2, s2 = measure(many, False, 0.25 * t_budget)
    out['also']['port_all_cores'] = {'value': v2, 'unit': 'rounds/s', 'cores': many, 'kind': 'port', 'sample': s2,
                                     'note': 'OPENBLAS threads = %d; only sdot inside np.linalg.norm is threaded' % many}
    if not isinstance(wl, AttackOnly):
        v3, s3 = measure(1, True, 0.3 * t_budget)
        out['also']['as_shipped'] = {'value': v3, 'unit': 'rounds/s', 'cores': 1, 'kind': 'port', 'sample': s3,
                                     'note': "oracle.shipped: the reference's own loops (dict of dicts, sorted(), Python "
                                             'sum, sorted(key=abs)), restated because /root/reference is not on this box'}
    return out


# ---- the line the driver parses: compact, bounded, tested ----------------------------------------------------------------
LINE_BUDGET = 4000         # bytes; the driver keeps an 8 KB tail of stdout + stderr (a 20 KB line was cut, parsed: null)


def sig(x, digits=5):
    """A float with `digits` significant digits (the record's numbers are measurements: 5 digits lose nothing)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float('%.*g' % (digits, float(x)))
    except (TypeError, ValueError):
        return x


def clip(text, limit):
    text = str(text)
    return text if len(text) <= limit else text[:limit - 3] + '...'


def compact_roofline(r):
    if not r:
        return None
    out = {k: sig(r.get(k)) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms',
                                       'launches_per_step', 'arithmetic') if k in r}
    if r.get('mfma_issued'):
        out['mfma_pipe_frac'] = sig(r['mfma_issued']['frac'])
    if r.get('with_slab_update'):
        out['frac_with_slab_update'] = sig(r['with_slab_update']['frac'], 4)
    if r.get('with_plane_split'):
        out['plane_split_ms'] = sig(r['with_plane_split']['plane_split_ms_per_step'])
        out['frac_with_split_and_update'] = sig(r['with_plane_split'].get('frac'), 4)
    return out


def compact_cpu_baseline(c):
    if not c:
        return None
    out = {'value': sig(c.get('value')), 'unit': c.get('unit'), 'cores': c.get('cores'), 'kind': c.get('kind'),
           'sample': clip(c.get('sample', ''), 260), 'host_cores': c.get('host_cores_available')}
    also = c.get('also') or {}
    for key, short in (('port_all_cores', 'all_cores'), ('as_shipped', 'as_shipped')):
        if key in also:
            out[short] = {'value': sig(also[key].get('value')), 'cores': also[key].get('cores')}
    return out


def compact_leg(rec):
    """One side leg as a handful of numbers: rate, time, the dominant kernel's roofline fraction."""
    if not rec:
        return None
    out = {'value': sig(rec.get('value')), 'ms': sig(rec.get('ms_per_step'))}      # ms: wall time of a whole step
    r = rec.get('roofline')
    if r:
        out['kernel'], out['bound'], out['frac'] = r.get('kernel'), r.get('bound'), sig(r.get('frac'), 4)
        # `frac` is the dominant kernel's algorithmic work over ITS OWN time (HIP events around its launches), not over `ms`:
        # the kernel time per step stands next to it so that the two can be told apart: frac over the wall
        # clock of a step is frac * kernel_ms / ms.  (For rounds of tens of microseconds the two come from different passes --
        # throughput without per-launch events, kernel time with them -- so kernel_ms may exceed ms by the events' overhead.)
        if r.get('avg_launch_ms') is not None and r.get('launches_per_step'):
            out['kernel_ms'] = sig(r['avg_launch_ms'] * r['launches_per_step'], 4)
    return out


def compact_line(detail, budget=LINE_BUDGET):
    """The ONE line of stdout: the contract's keys, the roofline and cpu_baseline objects, a flat kernel table, the
    north-star legs and one number per side workload.  Everything else lives in the detail file.  Optional sections are
    dropped, least important first, until the line fits the budget; the contract's own keys never are."""
    line = {k: detail.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'ranks_seen', 'steps', 'warmup', 'ms_per_step',
                                       'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')}
    line['value'], line['ms_per_step'] = sig(line['value'], 6), sig(line['ms_per_step'], 6)
    line['dtype'] = clip(line['dtype'], 120)
    if detail.get('rehearsal'):
        line['rehearsal'] = detail['rehearsal']
    cfg = detail.get('config') or {}
    line['config'] = {k: (clip(v, 220) if isinstance(v, str) else v) for k, v in cfg.items()
                      if k in ('workload', 'clients', 'params', 'corrupted', 'layout', 'params_per_gpu')}
    line['roofline'] = compact_roofline(detail.get('roofline'))
    line['cpu_baseline'] = compact_cpu_baseline(detail.get('cpu_baseline'))
    if detail.get('kernels'):
        line['kernels_ms'] = {k: sig(v['ms_per_step'], 4) for k, v in detail['kernels'].items()}
    others = detail.get('other_workloads') or {}
    star = {}
    for name in ('c5s', 'c5u'):
        rec = others.get(name)
        if rec:
            leg = compact_leg(rec)
            p8 = (rec.get('projected') or {}).get('gpus_8')
            if p8:
                leg['projected_8'] = sig(p8['rounds_per_s'], 4)
            leg['loop_ms'] = sig((rec.get('kernels') or {}).get('bulyan_loop', {}).get('ms_per_step'), 4)
            star[name] = leg
    if star:
        if 'c5u' in star:
            # north_star: >= 1 round/s at N = 10,000 x D = 25M on eight GPUs; said plainly (projection: one GPU is all there is)
            star['c5u_target_met'] = bool((star['c5u'].get('projected_8') or 0.0) >= 1.0)
        if 'c5s' in star:
            star['c5s_target_met'] = bool((star['c5s'].get('projected_8') or 0.0) >= 1.0)
        star['note'] = 'one GPU slice of configs[4] (D = 25M / 8); projected_8 = measured kernels + xGMI model, NOT measured on 8 GPUs'
        line['north_star'] = star
    side = {k: {kk: vv for kk, vv in compact_leg(v).items() if kk != 'kernel'} for k, v in others.items()
            if k not in ('c5s', 'c5u')}
    if side:
        line['others'] = side
    proj = detail.get('projected')
    if proj:
        line['projected'] = {k[5:]: sig(v['rounds_per_s'], 4) for k, v in proj.items() if k.startswith('gpus_')}
        line['projected']['note'] = 'PROJECTED (kernels measured here + xGMI link model), not measured'
    w1 = detail.get('sharded_path_w1') or {}
    if w1:
        # every leg with the columns it ran on: the clients leg runs on a QUARTER of the matrix at one rank (its config says
        # why), so its time is also given scaled to the headline's column count
        w1_line = {}
        for k, v in w1.items():
            if not isinstance(v, dict):
                continue
            cols = (v.get('config') or {}).get('params')
            w1_line[k] = {'ms': sig(v.get('ms_per_step'), 5), 'params': cols}
            full = cfg.get('params')
            if cols and full and cols != full and v.get('ms_per_step'):
                w1_line[k]['ms_scaled_to_%d_params' % full] = sig(v['ms_per_step'] * full / cols, 5)
        line['sharded_path_w1_ms'] = w1_line
    if detail.get('other_layout'):
        leg = detail['other_layout']
        line['other_layout'] = ({'layout': leg.get('layout'), 'error': clip(leg['error'], 160)} if leg.get('error')
                                else dict(compact_leg(leg), layout=leg.get('layout'), steps=leg.get('steps')))
    if detail.get('collectives'):
        line['collectives_ms'] = {k: sig(v['ms_per_step'], 4) for k, v in detail['collectives'].items()}
    if detail.get('verified_after_timing'):
        line['verified_after_timing'] = detail['verified_after_timing']
    if detail.get('detail_file'):
        line['detail_file'] = detail['detail_file']
    for victim in ('collectives_ms', 'sharded_path_w1_ms', 'others', 'projected', 'other_layout', 'kernels_ms',
                   'verified_after_timing', 'north_star'):
        if len(json.dumps(line)) <= budget:
            break
        line.pop(victim, None)
    text = json.dumps(line)
    if len(text) > budget:      # the contract's keys alone overflow: shorten the free-text fields, never drop a key
        line['config'] = {'workload': clip(cfg.get('workload', ''), 100)}
        if line.get('cpu_baseline'):
            line['cpu_baseline']['sample'] = clip(line['cpu_baseline']['sample'], 80)
        line['dtype'] = clip(line['dtype'], 40)
        text = json.dumps(line)
    assert len(text) <= budget and '\n' not in text, 'bench: the stdout line is %d bytes' % len(text)
    return text


def emit(detail, path):
    """Full record -> the detail file (best effort: a read-only tree must not cost the run its line); compact line -> the
    last line of stdout.  Nothing of the record goes to stderr: the driver's tail is stdout + stderr together."""
    if path:
        try:
            with open(path, 'w') as fh:
                json.dump(detail, fh, indent=1)
            detail['detail_file'] = path
        except OSError as exc:
            sys.stderr.write('bench: detail file not written (%s)\n' % exc)
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)     # RCCL's banner goes through C stdio: out first, the JSON line stays last
    except OSError:
        pass
    sys.stderr.flush()
    print(compact_line(detail), flush=True)


# ---- main -----------------------------------------------------------------------------------------
def collectives_table(agg, steps):
    """The timed steps' collectives: bytes THIS rank received and the time its compute stream spent in (or, for the overlapped
    gathers, waiting for) each of them, per step."""
    return {k: {'calls_per_step': v['calls'] / steps, 'MB_per_step': v['bytes'] / steps / 1e6, 'ms_per_step': v['ms'] / steps}
            for k, v in agg.comm_report().items()}


def bulyan_record(torch, dist, wl, eng, agg, steps, warmup, world, traffic):
    """One timed leg of a BulyanSharded workload as a self-contained record (value, roofline, kernels, collectives)."""
    elapsed, per_kernel = timed_steps(torch, dist, wl, eng, steps, warmup, world, agg=agg)
    rec = {'config': wl.config(), 'value': steps / elapsed, 'unit': 'rounds/s', 'ms_per_step': elapsed / steps * 1e3,
           'steps': steps, 'warmup': warmup, 'roofline': roofline_of(wl, per_kernel, traffic, steps),
           'kernels': kernel_table(per_kernel, steps), 'collectives': collectives_table(agg, steps),
           'verified_after_timing': wl.verify()}
    return rec


def main(argv=None):
    args = parse_args(argv)
    own_argv = sys.argv[1:] if argv is None else list(argv)
    if args.workload == 'launch-selftest':     # CPU / gloo: the launcher under test, no GPU work
        plan, why = launch_plan(args.gpus, os.environ, 0, needs_gpu=False)
        if plan == 'fail':
            raise SystemExit('bench: ' + why)
        if plan == 'spawn':
            raise SystemExit(spawn_ranks(args.gpus, own_argv))
        return launch_selftest(args)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the aggregation path has no CPU implementation')
    # BYZ_BENCH_ONE_DEVICE=1: a REHEARSAL of the N-rank run on a box with one GPU -- every rank on GPU 0, the collectives over gloo
    # (RCCL refuses two ranks on one device).  It exercises everything of the W > 1 path but RCCL itself (the launcher, the rank
    # count, the barriers and the max-over-ranks clock, the sharded workload, the side leg and its watchdog); its line says
    # `rehearsal` and is NOT a multi-GPU measurement.
    one_device = os.environ.get('BYZ_BENCH_ONE_DEVICE') == '1'
    plan, why = launch_plan(args.gpus, os.environ, args.gpus if one_device else torch.cuda.device_count())
    if plan == 'fail':
        raise SystemExit('bench: ' + why)
    if plan == 'spawn':
        raise SystemExit(spawn_ranks(args.gpus, own_argv))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if one_device else int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    os.environ.setdefault('BYZ_DEVICE', str(local_rank))
    if world > 1 or os.environ.get('BYZ_FORCE_COLLECTIVES') == '1':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if one_device:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)
    # n_gpus is what the collective library says, never what the command line says
    n_gpus = dist.get_world_size() if dist.is_initialized() else 1
    seen = ranks_seen(torch, dist, device)
    if n_gpus != args.gpus or seen != args.gpus:
        raise SystemExit('bench: --gpus %d but %d rank(s) joined the process group (%d counted)' % (args.gpus, n_gpus, seen))

    from attacking_federate_learning_amd.engine import Engine
    from attacking_federate_learning_amd.sharded import HipKernels, ShardedAggregator
    eng = Engine(local_rank)
    agg = ShardedAggregator(HipKernels(eng))
    traffic = load_traffic_table()

    if args.workload in ('c4', 'c5s', 'c5u'):
        n = args.clients or (4000 if args.workload == 'c4' else 10000)
        # c5s / c5u: the slice of configs[4] one GPU of eight would hold (25M/8 columns) -- 125 GB
        d_total = args.params or (10_000_000 if args.workload == 'c4' else 3_125_000 * world)
        primary = 'columns' if args.layout == 'both' else args.layout
        wl = BulyanSharded(torch, agg, eng, n, d_total, device, 1237, with_attack=args.workload != 'c4', layout=primary,
                           distinct=args.workload == 'c5u')
    elif args.workload == 'c3':
        wl = TrimmedMeanC3(torch, eng, args.clients or 1000, args.params or 1_000_000, device, 1236)
    elif args.workload == 'c2':
        wl = KrumC2(torch, eng, args.clients or 100, args.params or 79510, device, 1235)
    else:
        wl = AttackOnly(torch, eng, args.clients or 2400, args.params or 4_000_000, device, 1238)

    elapsed, per_kernel = timed_steps(torch, dist, wl, eng, args.steps, args.warmup, world, agg=agg)
    ms_per_step = elapsed / args.steps * 1e3
    line = {
        'metric': 'aggregation rounds/sec at N clients x D params (%s)' % wl.defence,
        'value': args.steps / elapsed, 'unit': 'rounds/s', 'n_gpus': n_gpus, 'ranks_seen': seen, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak' if args.workload in ('c5s', 'c5u') else 'strong', 'vs_baseline': None, 'dtype': wl.dtype(),
        'data': 'synthetic', 'config': wl.config(),
        'roofline': roofline_of(wl, per_kernel, traffic, args.steps),
        'kernels': kernel_table(per_kernel, args.steps),
        'collectives': collectives_table(agg, args.steps),
    }
    if one_device:
        line['rehearsal'] = ('BYZ_BENCH_ONE_DEVICE=1: %d rank(s) share GPU 0, collectives over gloo -- a rehearsal of the launch '
                             'path, NOT a multi-GPU measurement' % world)
    if hasattr(wl, 'verify'):
        line['verified_after_timing'] = wl.verify()
    if world == 1 and isinstance(wl, BulyanSharded) and wl.layout == 'columns':
        line['projected'] = projected_scaling(wl, line['kernels'], ms_per_step)
    want_side_leg = args.workload in ('c4', 'c5s', 'c5u') and (args.layout == 'both' or (
        world > 1 and args.layout == 'columns' and os.environ.get('BYZ_BENCH_ONE_LAYOUT') != '1'))

    def side_leg():
        """The other layout, a few steps (the clients layout moves 7/8 of G per round): north_star names client sharding with
        row tiles exchanged over xGMI; which layout is faster is a measurement, so every multi-GPU run
        records both.  Whatever happens in it is reported, never raised."""
        other_name = 'clients' if wl.layout == 'columns' else 'columns'
        wl.g = None
        torch.cuda.empty_cache()
        side_steps, side_warmup = min(args.steps, 3), min(args.warmup, 1)
        try:
            other = BulyanSharded(torch, agg, eng, n, d_total, device, 1237, with_attack=args.workload != 'c4', layout=other_name,
                                  distinct=args.workload == 'c5u')
            rec = bulyan_record(torch, dist, other, eng, agg, side_steps, side_warmup, world, None)
            rec['layout'] = other_name
            return rec, other
        except Exception as exc:      # noqa: BLE001
            return {'layout': other_name, 'error': ('%s: %s' % (type(exc).__name__, exc))[:300]}, None

    if want_side_leg and world > 1:
        # At W > 1 the headline's line goes out FIRST: the side leg runs code no multi-GPU box has run yet (one GPU is all the
        # builder ever had), and neither an exception nor a hang in it may cost the record its headline.  A watchdog ends every
        # rank with status 0 if the leg outlives its budget; its result, when there is one, goes to the detail file.
        if rank == 0:
            emit(line, args.detail_file)
        import threading
        def give_up():
            # the headline is out already; a hang of the side leg must not cost it -- but must not look like a clean finish
            # either: the detail file gets a marker, stderr a note, and then every rank ends
            if rank == 0 and args.detail_file:
                try:
                    line['other_layout'] = {'side_leg': 'timeout', 'layout': 'clients' if wl.layout == 'columns' else 'columns',
                                            'seconds': float(os.environ.get('BYZ_BENCH_SIDE_LEG_SECONDS', '150'))}
                    with open(args.detail_file, 'w') as fh:
                        json.dump(line, fh, indent=1)
                except (OSError, TypeError, ValueError):
                    pass
            try:
                os.write(2, b'bench.py: the side leg (the other layout) outlived its budget; ended by the watchdog\n')
            except OSError:
                pass
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get('BYZ_BENCH_SIDE_LEG_SECONDS', '150')), give_up)
        watchdog.daemon = True
        watchdog.start()
        rec, _ = side_leg()
        watchdog.cancel()
        if rank == 0 and args.detail_file:
            line['other_layout'] = rec
            try:
                with open(args.detail_file, 'w') as fh:
                    json.dump(line, fh, indent=1)
            except OSError:
                pass
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    if want_side_leg:
        line['other_layout'], other = side_leg()
        if other is not None:
            wl = other        # (the legs below run on the matrix that is resident now)

    if rank == 0 and world == 1:
        if args.workload == 'c4' and not args.no_sharded_w1 and not dist.is_initialized() and getattr(wl, 'g', None) is not None:
            line['sharded_path_w1'] = sharded_path_at_one_rank(torch, dist, wl, eng, device, traffic)
            if wl.layout == 'columns' and 'columns' in line['sharded_path_w1']:
                # the projection is built from the kernels that run at W > 1 (Gram -> all-reduce -> distances + near pairs)
                rec = line['sharded_path_w1']['columns']
                booked = sum(v['ms_per_step'] for v in rec['collectives'].values())
                line['projected'] = projected_scaling(wl, rec['kernels'], rec['ms_per_step'] - booked)
                line['projected']['built_from'] = 'sharded_path_w1.columns (collectives forced at one rank; their time at W = 1 subtracted)'
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(wl, args.cpu_seconds)
        if not args.no_extras and args.workload == 'c4':
            wl.g = None
            torch.cuda.empty_cache()
            extras = {}
            for make in (lambda: TrimmedMeanC3(torch, eng, 1000, 1_000_000, device, 1236),
                         lambda: KrumC2(torch, eng, 100, 79510, device, 1235),
                         lambda: KrumC2(torch, eng, 100, 21840, device, 1234),
                         lambda: AttackOnly(torch, eng, 2400, 1_000_000, device, 1238),
                         lambda: NoDefense(torch, eng, 1000, 1_000_000, device, 1243),
                         lambda: ServerUpdate(torch, eng, 10_000_000, device, 1244),
                         # the steps either side of the path; MnistNet's parameter shapes
                         lambda: BackdoorHook(torch, eng, 10_000_000, device, 1239),
                         lambda: Assembly(torch, eng, 100, [(100, 784), (100,), (10, 100), (10,)], device, 1240),
                         lambda: Assembly(torch, eng, 100, [(100, 784), (100,), (10, 100), (10,)], device, 1241, batched=True),
                         lambda: ClientStep(torch, eng, 100, 83, device, 1242)):
                w2 = make()
                # These rounds are tens of microseconds to a millisecond long: ten of them after three warm-up rounds measure
                # the chip on its way up from idle, not the round (round 6, same box: configs[2]'s trimmed mean 1.21-1.23 ms per
                # round over 10 steps, 1.12 ms over 200 -- the KERNEL is that much slower in the short run,
                # profiles/r06j_short_run_artifact.txt).  So every side workload is timed over at least ~0.25 s of rounds
                # behind a warm-up of a tenth of that (--extras-steps is the minimum).  The per-launch HIP events are a
                # visible share of such rounds: the throughput comes from a pass without them, the per-kernel table from a pass
                # with them.
                k2, warm2 = side_leg_steps(timed_steps(torch, dist, w2, eng, 3, 2, 1, events=False)[0] / 3.0, args.extras_steps)
                e2, _ = timed_steps(torch, dist, w2, eng, k2, warm2, 1, events=False)
                k_ev = min(k2, 500)
                _, pk2 = timed_steps(torch, dist, w2, eng, k_ev, max(warm2 // 4, 1), 1)
                key = '%s_D%d' % (w2.name, w2.d)
                if key in extras:
                    key += '_batched'
                extras[key] = {'config': w2.config(), 'value': k2 / e2, 'unit': 'rounds/s', 'steps': k2, 'warmup': warm2,
                               'ms_per_step': e2 / k2 * 1e3, 'roofline': roofline_of(w2, pk2, traffic, k_ev),
                               'kernels': kernel_table(pk2, k_ev)}
                del w2
                torch.cuda.empty_cache()
            if not args.no_north_star:
                # BASELINE configs[4]'s slice of one GPU of eight (N = 10,000, D = 25M / 8), with and without the
                # identical-row shortcut: c5s = the attack as the reference runs it (its m rows are ONE vector, the Gram runs
                # over the N - m + 1 unique rows); c5u = 10,000 DISTINCT rows (an attacker who adds per-row noise: nothing to
                # fold), the full N^2 D Gram.  Both with the 8-GPU projection.
                shared = None      # c5u leaves the matrix as generated (no write-back), so c5s can run on the same 125 GB after it
                for name in ('c5u', 'c5s'):
                    w5 = BulyanSharded(torch, agg, eng, 10000, 3_125_000, device, 1237, with_attack=True, layout='columns',
                                       distinct=name == 'c5u', g=shared)
                    shared = w5.g
                    rec = bulyan_record(torch, dist, w5, eng, agg, max(args.north_star_steps, 1), 1, 1, traffic)
                    rec['projected'] = projected_scaling(w5, rec['kernels'], rec['ms_per_step'])
                    extras[name] = rec
                    w5.g = None
                    del w5
                del shared
                torch.cuda.empty_cache()
            line['other_workloads'] = extras
    if rank == 0:
        emit(line, args.detail_file)
    if dist.is_initialized():
        dist.destroy_process_group()


def sharded_path_at_one_rank(torch, dist, wl, eng, device, traffic, steps=2, warmup=1):
    """The code path that runs at W > 1, timed on the one GPU there is: BYZ_FORCE_COLLECTIVES=1 makes ShardedAggregator issue
    every collective through RCCL at world size 1, so the columns layout goes Gram -> all-reduce -> distances_from_gram +
    near-pair exchange (not the engine's one-call composition the headline uses at W = 1), and the clients layout its
    per-panel gather + tile share + N x N accumulation.  Same matrix as the headline (wl.g), few steps."""
    from attacking_federate_learning_amd.sharded import HipKernels, ShardedAggregator
    os.environ['BYZ_FORCE_COLLECTIVES'] = '1'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ['MASTER_PORT'] = str(free_port())
    os.environ['RANK'], os.environ['WORLD_SIZE'] = '0', '1'
    out = {'note': 'BYZ_FORCE_COLLECTIVES=1 at world size 1: the W > 1 code path of sharded.py with every collective issued '
                   'through RCCL (send-to-self); %d timed steps per layout' % steps}
    # RCCL prints its NCCL_DEBUG=VERSION banner through C stdio on fd 1; the contract's stdout is ONE JSON line, so fd 1 points
    # at stderr while the process group of this leg lives
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    dist.init_process_group('nccl', device_id=device)
    try:
        forced = ShardedAggregator(HipKernels(eng))
        assert forced.always_collective
        for layout in ('columns', 'clients'):
            # clients at one rank re-shards the theta selected rows through a send-to-self: a second and third copy of
            # theta x D floats next to the 160 GB matrix.  That leg therefore runs on the first quarter of the columns (a
            # strided view of the same matrix: 40 GB), and says so in its config.
            cols = wl.d_total if layout == 'columns' else wl.d_total // 4
            w = BulyanSharded(torch, forced, eng, wl.n, cols, device, 1237, with_attack=wl.with_attack, layout=layout,
                              distinct=wl.distinct, g=wl.g if cols == wl.d_total else wl.g[:, :cols])
            rec = bulyan_record(torch, dist, w, eng, forced, steps, warmup, 1, None)
            rec['layout'] = layout
            out[layout] = rec
            w.g = None
            torch.cuda.empty_cache()
    finally:
        dist.destroy_process_group()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
        for key in ('BYZ_FORCE_COLLECTIVES', 'RANK', 'WORLD_SIZE', 'MASTER_PORT'):
            os.environ.pop(key, None)
    return out


if __name__ == '__main__':
    main()
