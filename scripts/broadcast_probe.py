"""The attack's write-back (one vector into m rows, malicious.py:26-27) at configs[4]'s slice: time per BYZ_BROADCAST_RUN (torch-free)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Engine
eng = Engine(0)
m, d = int(sys.argv[1]) if len(sys.argv) > 1 else 2400, int(sys.argv[2]) if len(sys.argv) > 2 else 3125000
g = eng.empty((m, d), np.float32)
small = eng.to_device(np.random.default_rng(0).standard_normal((8, d), dtype=np.float32))
import ctypes
from attacking_federate_learning_amd.engine import _vp, _check
# seed the first 8 rows so that the statistics are finite, then time drift_attack with write_back over all m rows
_check(eng.lib.byz_upload(eng.ctx, _vp(g.ptr), small.numpy().ctypes.data_as(ctypes.c_void_p), 8 * d * 4, None))
for run in (1, 4, 8, 16, 8, 1):
    os.environ['BYZ_BROADCAST_RUN'] = str(run)
    outs = [eng.empty((d,), np.float32) for _ in range(3)]
    def one():
        _check(eng.lib.byz_drift_attack_dev(eng.ctx, _vp(g.ptr), m, d, d, 1.5, _vp(outs[0].ptr), _vp(outs[1].ptr), _vp(outs[2].ptr), 1, None))
    one(); eng.synchronize()
    eng.timing(True)
    for _ in range(3):
        one()
    eng.synchronize()
    t = eng.timing_read(); eng.timing(False)
    print('run=%d: misc (write-back and its two small neighbours) %.3f ms per call (%d launches), column_stats %.3f ms' % (
        run, t['misc']['total_ms'] / 3, t['misc']['launches'] // 3, t['column_stats']['total_ms'] / t['column_stats']['launches']), flush=True)
row = g.numpy()[[0, 1, m // 2, m - 1]]
assert np.array_equal(row[0], row[1]) and np.array_equal(row[0], row[3]) and np.array_equal(row[0], outs[0].numpy())
print('rows identical to the drift vector: ok')
