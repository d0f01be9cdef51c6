import os, sys, time
import numpy as np
ROOT = '/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from attacking_federate_learning_amd.engine import Engine, Distances
from test_gpu_scale import point_distances
from oracle import scale
eng = Engine(0)
for n in (4000, 10000):
    f = int(n * 0.24)
    d = point_distances(4100 + n, n)
    dev = Distances(eng.to_device(d), n)
    idx = eng.krum_select(dev, n, f)
    eng.timing(True)
    for _ in range(3):
        idx = eng.krum_select(dev, n, f)
    t = eng.timing_read(); eng.timing(False)
    print('N=%d krum index %d (oracle %d); row_sort %.3f ms per call' % (n, idx, scale.krum_pick(d, n, f), t['row_sort']['total_ms'] / 3), flush=True)
