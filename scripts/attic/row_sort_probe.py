"""Row sort with tables (the Bulyan loop's preparation) and the loop itself at N = 4000 / 10,000 (torch-free GPU probe):
per-kernel times and the selection against the C oracle's sampled picks."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from attacking_federate_learning_amd.engine import Engine, Distances
from test_gpu_scale import point_distances, check_selection
eng = Engine(0)
for n in (4000, 10000):
    f = int(n * 0.24)
    d = point_distances(4100 + n, n)
    dev = Distances(eng.to_device(d), n)
    sel = eng.bulyan_select(dev, n, f)
    eng.timing(True)
    for _ in range(3):
        sel = eng.bulyan_select(dev, n, f)
    t = eng.timing_read(); eng.timing(False)
    print('N=%d: row_sort %.3f ms, loop %.2f ms per call' % (n, t['row_sort']['total_ms'] / 3, t['bulyan_loop']['total_ms'] / 3), flush=True)
    check_selection(d, n, f, np.asarray(sel).tolist())
    print('   selection = that of the oracle', flush=True)
