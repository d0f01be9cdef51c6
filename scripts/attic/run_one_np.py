"""Run one kernel family a few times, torch-free (profiling target).  usage: run_one_np.py tm N D [c]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Engine
eng = Engine(0)
what, n, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(os.environ.get('ITERS', '5'))
g = eng.to_device(np.random.default_rng(0).standard_normal((n, d), dtype=np.float32))
for _ in range(iters):
    if what == 'tm':
        out = eng.trimmed_mean(g, n, int(sys.argv[4]) if len(sys.argv) > 4 else n // 5)
eng.synchronize()
print('done', what, n, d, 'redone', eng.trimmed_mean_redone())
