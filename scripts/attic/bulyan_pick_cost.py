"""What one pick of the Bulyan loop costs with and without re-scoring, by the number of workgroups (256 rows each): the loop
kernel's time / theta for several N (torch-free).  BYZ_BULYAN_BAND=0 = fp64 decisions (no re-score: NOT the reference's selection)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from attacking_federate_learning_amd.engine import Engine, Distances
from test_gpu_scale import point_distances
eng = Engine(0)
for n in (200, 256, 400, 512, 1024, 2048, 4000):
    f = int(n * 0.24)
    dev = Distances(eng.to_device(point_distances(4100 + n, n)), n)
    for band in ('0', 'rigorous'):
        os.environ['BYZ_BULYAN_BAND'] = band
        eng.bulyan_select(dev, n, f)
        eng.timing(True)
        for _ in range(3):
            eng.bulyan_select(dev, n, f)
        t = eng.timing_read(); eng.timing(False)
        ms = t['bulyan_loop']['total_ms'] / 3
        print('N=%5d (%2d workgroups) band=%-8s: loop %.3f ms, theta %d, %.2f us per pick, re-scored %d' % (
            n, -(-n // 256), band, ms, n - 2 * f, 1e3 * ms / (n - 2 * f), eng.bulyan_rescored()), flush=True)
