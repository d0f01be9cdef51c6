"""First GPU visit of the next round (torch-free, ~40 s): the opt-in variants written without a GPU at the end of round 2.

  1. BYZ_BULYAN_RESCORE=plain (literal fp32 chain, four batches of table entries in flight) against the default:
     selections must be identical; time at N = 4000 and 10,000.
  2. BYZ_TM_HIST16=1 (16-bit histogram counters: two workgroups per CU for the 8-wave row-split trimmed mean) against the
     default at 2080 rows:
     results must agree to 1e-6; time and tiles handed to the general kernel.
  3. the branch-free staging loads of the <= 1024-row trimmed-mean kernel (default since the end of round 2, bitwise
     checked) against BYZ_TM_FETCH=guarded (sixteen loads per tile, one round trip at a time) at 1000 rows: the time.
  4. run scripts/small_krum_check.py next: it now also covers BYZ_KRUM_SMALL_TAIL=1 (K3..K5 in one launch), cases and timings.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from attacking_federate_learning_amd.engine import Engine, Distances   # noqa: E402
from test_gpu_scale import point_distances                             # noqa: E402


def main():
    eng = Engine(0)
    for n in (4000, 10000):
        f = int(n * 0.24)
        dev = Distances(eng.to_device(point_distances(4100 + n, n)), n)
        got = {}
        for mode in ('v1', 'plain'):
            os.environ['BYZ_BULYAN_RESCORE'] = mode
            eng.bulyan_select(dev, n, f)
            t0 = time.perf_counter()
            got[mode] = eng.bulyan_select(dev, n, f).tolist()
            print('bulyan N=%d %s: %.1f ms, re-scored %d' % (n, mode, 1e3 * (time.perf_counter() - t0), eng.bulyan_rescored()), flush=True)
        print('  selections', 'identical' if got['v1'] == got['plain'] else 'DIFFERENT', flush=True)
    os.environ.pop('BYZ_BULYAN_RESCORE', None)

    rows, cols, corrupted = 2080, 1 << 18, 1920
    rng = np.random.default_rng(5)
    g = rng.standard_normal((rows, cols), dtype=np.float32)
    buf = eng.to_device(g)
    out = {}
    for mode in ('0', '1'):
        os.environ['BYZ_TM_HIST16'] = mode
        o = eng.trimmed_mean(buf, rows, corrupted)
        eng.synchronize()
        t0 = time.perf_counter()
        outs = [eng.trimmed_mean(buf, rows, corrupted) for _ in range(20)]
        eng.synchronize()
        dt = (time.perf_counter() - t0) / 20
        out[mode] = o.numpy()
        print('trimmed mean %d x %d hist16=%s: %.3f ms (%.2f TB/s), tiles redone %d' % (
            rows, cols, mode, dt * 1e3, 4.0 * rows * cols / dt / 1e12, eng.trimmed_mean_redone()), flush=True)
        del outs
    print('  max |difference| between the two: %.3e' % np.abs(out['0'] - out['1']).max(), flush=True)
    os.environ.pop('BYZ_TM_HIST16', None)

    rows, cols, corrupted = 1000, 1 << 18, 200
    g = rng.standard_normal((rows, cols), dtype=np.float32)
    buf = eng.to_device(g)
    out = {}
    for mode in ('default', 'bf'):     # 'default' here = the former guarded staging, 'bf' = the branch-free one (now the default)
        if mode == 'default':
            os.environ['BYZ_TM_FETCH'] = 'guarded'
        else:
            os.environ.pop('BYZ_TM_FETCH', None)
        o = eng.trimmed_mean(buf, rows, corrupted)
        eng.synchronize()
        t0 = time.perf_counter()
        outs = [eng.trimmed_mean(buf, rows, corrupted) for _ in range(40)]
        eng.synchronize()
        dt = (time.perf_counter() - t0) / 40
        out[mode] = o.numpy()
        print('trimmed mean %d x %d staging %s: %.3f ms (%.2f TB/s), tiles redone %d' % (
            rows, cols, mode, dt * 1e3, 4.0 * rows * cols / dt / 1e12, eng.trimmed_mean_redone()), flush=True)
        del outs
    print('  bitwise equal:', bool(np.array_equal(out['default'], out['bf'], equal_nan=True)), flush=True)


if __name__ == '__main__':
    main()
