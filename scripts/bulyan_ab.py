"""Same-box A/B of the Bulyan loop's re-score (BYZ_BULYAN_RESCORE=plain | marked; torch-free GPU probe): time of the whole loop
(row sorts included) and whether the selections agree pick for pick -- `plain` is the form the C oracle was checked against."""
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
    modes = sys.argv[1].split(',') if len(sys.argv) > 1 else ['plain', 'marked', 'coop']
    sizes = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [4000, 10000]
    eng = Engine(0)
    for n in sizes:
        f = int(n * 0.24)
        for identical in (0, f):      # the scaled family, and the attack's f identical rows
            dev = Distances(eng.to_device(point_distances(4100 + n, n, identical=identical)), n)
            ref = None
            for rep in range(2):
                for m in modes:
                    os.environ['BYZ_BULYAN_RESCORE'] = m
                    sel = eng.bulyan_select(dev, n, f)
                    eng.timing(True)
                    t0 = time.perf_counter()
                    sel = eng.bulyan_select(dev, n, f)
                    wall = 1e3 * (time.perf_counter() - t0)
                    t = eng.timing_read()
                    eng.timing(False)
                    sel = np.asarray(sel)
                    if ref is None:
                        ref = sel
                    print('N=%d identical=%d %-5s: loop kernel %.2f ms (wall %.1f ms), re-scored %d, same selection as the first: %s' % (
                        n, identical, m, t['bulyan_loop']['total_ms'], wall, eng.bulyan_rescored(), bool(np.array_equal(sel, ref))), flush=True)


if __name__ == '__main__':
    main()
