"""Minimal GPU check (torch-free, ~1 s): the branch-free staging of the <= 1024-row trimmed mean against the guarded one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Engine   # noqa: E402

eng = Engine(0)
rng = np.random.default_rng(3)
ok = True
for rows, cols, c in ((1000, 8200, 200), (300, 4099, 70), (1024, 4096, 1), (513, 1000, 500)):
    g = rng.standard_normal((rows, cols), dtype=np.float32)
    buf = eng.to_device(g)
    idx = rng.permutation(rows)[: rows - 7].astype(np.int32)
    res = {}
    for mode in ('default', 'bf'):
        if mode == 'default':
            os.environ['BYZ_TM_FETCH'] = 'guarded'
        else:
            os.environ.pop('BYZ_TM_FETCH', None)
        t0 = time.perf_counter()
        a = eng.trimmed_mean(buf, rows, c).numpy()
        b = eng.trimmed_mean(buf, rows, c, row_index=idx).numpy()
        res[mode] = (a, b, time.perf_counter() - t0)
    same = np.array_equal(res['default'][0], res['bf'][0], equal_nan=True) and np.array_equal(res['default'][1], res['bf'][1], equal_nan=True)
    ok = ok and same
    print(rows, cols, c, 'bitwise equal' if same else 'DIFFERENT', flush=True)
print('ALL EQUAL' if ok else 'MISMATCH', flush=True)
