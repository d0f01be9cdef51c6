"""What the kernels beyond 16,384 rows (csrc/large_rows.hip) cost: wall time of Krum's selection, Bulyan's selection and the trimmed
mean at 16,500 / 20,000 / 32,000 rows, and of the SAME calls through the forced large path at configs[3]'s and configs[4]'s row counts
next to the production kernels (what "built to be there, not to be fast" amounts to).  Needs an MI355X.

    python scripts/large_rows_timing.py [--quick]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Distances, get_engine  # noqa: E402


def point_distances(seed, n, dim, identical=0):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    pts *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if identical:
        pts[:identical] = pts[:identical].mean(axis=0)
    p = pts.astype(np.float64 if n <= 20000 else np.float32)      # (three n x n temporaries: 8 GB each in fp64 at n = 32,000)
    sq = (p * p).sum(1)
    d = sq[:, None] + sq[None, :]
    d -= 2.0 * (p @ p.T)
    d = np.sqrt(np.maximum(d, 0.0)).astype(np.float32)
    d = np.minimum(d, d.T)
    if identical:
        d[:identical, :identical] = 0.0
    np.fill_diagonal(d, np.inf)
    return d


def timed(eng, fn, repeat=2):
    best = float('inf')
    for _ in range(repeat):
        eng.synchronize()
        t0 = time.perf_counter()
        out = fn()
        eng.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


def main():
    quick = '--quick' in sys.argv
    eng = get_engine()
    sizes = [(4000, 0), (10000, 0), (16500, 0), (20000, 0), (20000, 4800)] + ([] if quick else [(32000, 0)])
    print('%-26s %12s %12s %12s   %s' % ('rows (identical)', 'krum ms', 'bulyan ms', 'rows re-scored', 'path'))
    for n, identical in sizes:
        f = int(0.24 * n)
        dist = Distances(eng.to_device(point_distances(100 + n, n, 16, identical)), n)
        for forced in ((False, True) if n <= 16384 else (None,)):
            if forced:
                os.environ['BYZ_SELECT_LARGE'] = '1'
            k_ms, _ = timed(eng, lambda: eng.krum_select(dist, n, f))
            b_ms, _ = timed(eng, lambda: eng.bulyan_select(dist, n, f, on_device=True), repeat=1 if n > 16384 else 2)
            print('%-26s %12.2f %12.1f %12d   %s' % ('%d (%d)' % (n, identical), k_ms, b_ms, eng.bulyan_rescored(),
                                                   'large_rows.hip' + (' (forced)' if forced else '') if forced is not False else 'select.hip'))
            os.environ.pop('BYZ_SELECT_LARGE', None)
        del dist
    print()
    print('%-26s %12s %12s   %s' % ('trimmed mean rows x cols', 'ms', 'GB/s', 'path'))
    rng = np.random.default_rng(5)
    for n, cols in [(5200, 65536), (16384, 16384), (16385, 16384), (20001, 16384)] + ([] if quick else [(40000, 16384)]):
        g = eng.to_device(rng.standard_normal((n, cols), dtype=np.float32))
        c = int(0.48 * n)
        for forced in ((False, True) if n <= 16384 else (None,)):
            if forced:
                os.environ['BYZ_TM_LARGE'] = '1'
            ms, _ = timed(eng, lambda: eng.trimmed_mean(g, n, c))
            print('%-26s %12.2f %12.1f   %s' % ('%d x %d' % (n, cols), ms, 4.0 * n * cols / ms / 1e6,
                                                'large_rows.hip' + (' (forced)' if forced else '') if forced is not False else 'trimmed_mean.hip / window_lean.hip'))
            os.environ.pop('BYZ_TM_LARGE', None)
        del g


if __name__ == '__main__':
    main()
