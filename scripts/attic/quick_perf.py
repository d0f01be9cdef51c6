"""Quick device-resident timings (development aid, not the contract bench)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from attacking_federate_learning_amd.engine import get_engine

eng = get_engine()

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

def report(name, ms, nbytes=None, flops=None):
    s = '%-40s %9.3f ms' % (name, ms)
    if nbytes: s += '  %7.1f GB/s (%.1f%% of 8 TB/s)' % (nbytes / ms / 1e6, nbytes / ms / 1e6 / 80)
    if flops: s += '  %7.2f TF' % (flops / ms / 1e9)
    print(s, flush=True)

which = sys.argv[1:] or ['c2', 'c3', 'attack', 'gram', 'bulyan']
gen = torch.Generator(device='cuda').manual_seed(0)
if 'c2' in which:
    for d in (21840, 79510):
        g = torch.randn((100, d), device='cuda', generator=gen)
        eng.timing(True)
        report('krum N=100 D=%d (index)' % d, timeit(lambda: eng.krum(g, 100, 24, return_index=True)), nbytes=4 * 100 * d)
        print('   ', {k: round(v['total_ms'] / v['launches'] * 1000, 1) for k, v in eng.timing_read().items()}, 'us/launch')
        eng.timing(False)
        report('krum N=100 D=%d (index, untimed)' % d, timeit(lambda: eng.krum(g, 100, 24, return_index=True)), nbytes=4 * 100 * d)
if 'c3' in which:
    for n, d in ((1000, 1 << 20), (1000, 1000000), (100, 1 << 22), (512, 1 << 20), (64, 1 << 22)):
        g = torch.randn((n, d), device='cuda', generator=gen)
        report('trimmed_mean N=%d D=%d' % (n, d), timeit(lambda: eng.trimmed_mean(g, n, n // 5), iters=5, warm=1), nbytes=4 * n * d)
        del g
if 'attack' in which:
    g = torch.randn((240, 1 << 22), device='cuda', generator=gen)
    report('drift_attack m=240 D=4M', timeit(lambda: eng.drift_attack(g, 1.5)), nbytes=4 * 240 * (1 << 22))
    report('no_defense  N=240 D=4M', timeit(lambda: eng.no_defense(g)), nbytes=4 * 240 * (1 << 22))
    del g
if 'gram' in which:
    for n, d in ((4000, 250000), (4000, 1000000)):
        g = torch.randn((n, d), device='cuda', generator=gen)
        report('distances N=%d D=%d' % (n, d), timeit(lambda: eng.pairwise_distances(g), iters=3, warm=1), nbytes=4 * n * d, flops=float(n) * n * d)
        del g
if 'bulyan' in which:
    for n, d in ((1000, 1 << 16), (4000, 1 << 14)):
        g = torch.randn((n, d), device='cuda', generator=gen)
        f = int(n * 0.24)
        eng.timing(True)
        report('bulyan N=%d D=%d f=%d' % (n, d, f), timeit(lambda: eng.bulyan(g, n, f), iters=2, warm=1))
        print('   ', {k: round(v['total_ms'] / v['launches'], 3) for k, v in eng.timing_read().items()}, 'ms/launch')
        eng.timing(False)
        del g
