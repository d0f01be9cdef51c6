"""GPU probe (torch-free): BYZ_BULYAN_RESCORE=plain (the re-score as a literal chain of fp32 additions) against the default."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from attacking_federate_learning_amd.engine import Engine, Distances   # noqa: E402
from test_gpu_scale import point_distances                             # noqa: E402
import numpy as np                                                     # noqa: E402


def quantised(seed, n):
    """Distances on a grid of 1/64: ties in almost every chunk."""
    d = point_distances(seed, n)
    q = np.round(d * 64.0) / 64.0
    q = np.where(np.isinf(d), d, q).astype(np.float32)
    return np.minimum(q, q.T)


eng = Engine(0)
bad = 0
for kind, n in (('points', 700), ('quantised', 900), ('points', 4000), ('points', 10000)):
    f = int(n * 0.24)
    if kind == 'points':
        dist = point_distances(4100 + n, n)
    elif kind == 'attack':
        dist = point_distances(4200 + n, n, identical=f)
    else:
        dist = quantised(4400 + n, n)
    dev = Distances(eng.to_device(dist), n)
    got = {}
    for mode in ('v1', 'plain'):
        os.environ['BYZ_BULYAN_RESCORE'] = mode
        eng.bulyan_select(dev, n, f)
        t0 = time.perf_counter()
        got[mode] = eng.bulyan_select(dev, n, f).tolist()
        dt = time.perf_counter() - t0
        print('%s N=%d %s: %.2f ms, re-scored %d' % (kind, n, mode, dt * 1e3, eng.bulyan_rescored()), flush=True)
    same = got['v1'] == got['plain']
    bad += 0 if same else 1
    print('  identical' if same else '  DIFFERENT at pick %d' % next(i for i, (a, b) in enumerate(zip(got['v1'], got['plain'])) if a != b), flush=True)
print('bad:', bad)
