"""Round-3 timing probe (torch-free): trimmed-mean layouts and shapes, the Bulyan loop.  Prints as it goes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from attacking_federate_learning_amd.engine import Engine, Distances   # noqa: E402


def tm(eng, rows, cols, corrupted, env, reps=30, label=''):
    saved = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    rng = np.random.default_rng(rows + cols)
    key = (rows, cols)
    if tm.cache.get('key') != key:
        tm.cache = {'key': key, 'buf': eng.to_device(rng.standard_normal((rows, cols), dtype=np.float32))}
    buf = tm.cache['buf']
    o = eng.trimmed_mean(buf, rows, corrupted)
    eng.synchronize()
    t0 = time.perf_counter()
    outs = [eng.trimmed_mean(buf, rows, corrupted) for _ in range(reps)]
    eng.synchronize()
    dt = (time.perf_counter() - t0) / reps
    res = o.numpy()
    print('trimmed mean %5d x %7d keep %4d %-28s %.3f ms  %.2f TB/s  frac %.3f  redone %d' % (
        rows, cols, rows - corrupted - 1, label or str(env), dt * 1e3, 4.0 * rows * cols / dt / 1e12,
        4.0 * rows * cols / dt / 8e12, eng.trimmed_mean_redone()), flush=True)
    del outs
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    return res


tm.cache = {}


def main():
    eng = Engine(0)
    what = sys.argv[1:] or ['tm', 'bulyan']
    if 'tm' in what:
        a = tm(eng, 1000, 1 << 18, 200, {'BYZ_TM_ROWS': None}, label='column layout (default)')
        b = tm(eng, 1000, 1 << 18, 200, {'BYZ_TM_ROWS': '1'}, label='row-split, 16-bit hist')
        c = tm(eng, 1000, 1 << 18, 200, {'BYZ_TM_ROWS': '1', 'BYZ_TM_HIST16': '0'}, label='row-split, 32-bit hist')
        print('  equal:', np.array_equal(a, b, equal_nan=True), np.array_equal(a, c, equal_nan=True), flush=True)
        tm(eng, 512, 1 << 18, 100, {'BYZ_TM_ROWS': None}, label='column layout (default)')
        tm(eng, 512, 1 << 18, 100, {'BYZ_TM_ROWS': '1'}, label='row-split, 16-bit hist')
        tm(eng, 2080, 1 << 18, 1920, {}, label='default (16-bit hist)')
        tm(eng, 2080, 1 << 18, 1920, {'BYZ_TM_HIST16': '0'}, label='32-bit hist')
        tm(eng, 2080, 1 << 18, 416, {}, label='default, keep 1663')
        tm(eng, 1536, 1 << 18, 300, {}, label='default')
        tm(eng, 5200, 1 << 17, 4800, {}, label='default (16 waves)')
        tm(eng, 5200, 1 << 17, 1040, {}, label='default (16 waves), keep 4159')
    if 'bulyan' in what:
        from test_gpu_scale import point_distances
        for n in (4000, 10000):
            f = int(n * 0.24)
            dev = Distances(eng.to_device(point_distances(4100 + n, n)), n)
            eng.bulyan_select(dev, n, f)
            t0 = time.perf_counter()
            eng.bulyan_select(dev, n, f)
            print('bulyan N=%d: %.1f ms, re-scored %d' % (n, 1e3 * (time.perf_counter() - t0), eng.bulyan_rescored()), flush=True)


if __name__ == '__main__':
    main()
