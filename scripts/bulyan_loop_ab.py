"""Same-box A/B of the Bulyan selection loop by environment settings (torch-free): the loop's kernel time on the distances of a
`scaled` matrix (N x 4096), alternated, and whether the selection is the first setting's.

    python scripts/bulyan_loop_ab.py 4000 BYZ_BULYAN_FRONT=0 BYZ_BULYAN_FRONT=1
    ATTACK=1: the first 0.24 N rows are one vector (the attack's exact ties)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd.engine import Engine   # noqa: E402


def main():
    n = int(sys.argv[1])
    settings = sys.argv[2:]
    f = int(n * 0.24)
    eng = Engine(0)
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n, 4096), dtype=np.float32)
    g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if os.environ.get('ATTACK'):
        g[:f] = g[:f].mean(axis=0)
    dist = eng.pairwise_distances(g)
    ref = None
    touched = set()
    for rep in range(int(os.environ.get('REPS', '3'))):
        for setting in settings:
            for key in touched:
                os.environ.pop(key, None)
            for kv in setting.split(','):
                key, val = kv.split('=', 1)
                os.environ[key] = val
                touched.add(key)
            sel = np.asarray(eng.bulyan_select(dist, n, f))
            eng.timing(True)
            for _ in range(2):
                sel = np.asarray(eng.bulyan_select(dist, n, f))
            t = eng.timing_read()
            eng.timing(False)
            if ref is None:
                ref = sel
            print('rep %d  %-28s bulyan_loop %8.3f ms  (%d rows re-scored)  selection equal: %s' % (
                rep, setting, t['bulyan_loop']['total_ms'] / 2, eng.bulyan_rescored(), bool(np.array_equal(sel, ref))), flush=True)


if __name__ == '__main__':
    main()
