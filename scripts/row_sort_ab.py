"""Same-box A/B of the selection kernels' row sort: the register-blocked bitonic network against the textbook form
(BYZ_ROW_SORT_BLOCKED=0), HIP-event time of the row_sort kernel per call (torch-free).  usage: row_sort_ab.py [N ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Engine
eng = Engine(0)
sizes = [int(a) for a in sys.argv[1:]] or [300, 1000, 2000, 4000, 7601, 10000]
for n in sizes:
    f = int(0.24 * n)
    g = np.random.default_rng(n).standard_normal((n, 96)).astype(np.float32)
    dist = eng.pairwise_distances(g)
    res = {'0': [], '1': []}
    sel = {}
    for rep in range(4):
        for mode in ('0', '1'):
            os.environ['BYZ_ROW_SORT_BLOCKED'] = mode
            eng.timing(True)
            for _ in range(3):
                s = eng.bulyan_select(dist, n, f)
            eng.synchronize()
            t = eng.timing_read()['row_sort']
            res[mode].append(t['total_ms'] / t['launches'])
            sel[mode] = list(s)
            eng.timing(False)
    assert sel['0'] == sel['1']
    print('N = %5d: textbook %s ms   blocked %s ms' % (n, ' '.join('%.3f' % x for x in res['0']), ' '.join('%.3f' % x for x in res['1'])), flush=True)
