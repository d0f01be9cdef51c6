"""Same-box A/B of the Bulyan loop's re-score (torch-free GPU probe): time of the whole loop and whether the selections agree
pick for pick.  `python scripts/bulyan_ab.py plain,marked 4000,10000` alternates BYZ_BULYAN_RESCORE (`plain` is the form the C
oracle was checked against); `python scripts/bulyan_ab.py BYZ_BULYAN_INCR=0,1 4000,10000` alternates any other variable."""
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
    var = 'BYZ_BULYAN_RESCORE'
    spec = sys.argv[1] if len(sys.argv) > 1 else 'plain,marked'
    settings = None
    if ';' in spec or spec.count('=') > 1:
        # "A=1,B=2;A=1,B=3": whole settings, alternated
        settings = [dict(kv.split('=', 1) for kv in one.split(',')) for one in spec.split(';')]
        modes = [','.join('%s=%s' % kv for kv in st.items()) for st in settings]
        var = 'settings'
    else:
        if '=' in spec:
            var, spec = spec.split('=', 1)
        modes = spec.split(',')
    sizes = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [4000, 10000]
    eng = Engine(0)
    for n in sizes:
        f = int(n * 0.24)
        for identical in ((0,) if os.environ.get('BYZ_AB_SCALED_ONLY') else (0, f)):      # the scaled family, and the attack's f identical rows
            dev = Distances(eng.to_device(point_distances(4100 + n, n, identical=identical)), n)
            ref = None
            for rep in range(1 if os.environ.get('BYZ_AB_SCALED_ONLY') else 2):
                for m in modes:
                    if settings is not None:
                        os.environ.update(settings[modes.index(m)])
                    else:
                        os.environ[var] = m
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
                    print('N=%d identical=%d %s=%-6s: loop kernel %.2f ms (wall %.1f ms), re-scored %d (%d from records), same selection as the first: %s' % (
                        n, identical, var, m, t['bulyan_loop']['total_ms'], wall, eng.bulyan_rescored(), eng.bulyan_from_records(),
                        bool(np.array_equal(sel, ref))), flush=True)


if __name__ == '__main__':
    main()
