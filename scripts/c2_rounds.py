"""configs[1] rounds back to back (torch-free): a profiling target for the N <= 128 Krum path.  usage: c2_rounds.py D [rounds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Engine
eng = Engine(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 79510
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(1)
g = rng.standard_normal((100, d), dtype=np.float32) * (1.0 + 0.5 * rng.permutation(100) / 100).astype(np.float32)[:, None]
buf = eng.to_device(g)
out = eng.empty((d,), np.float32)
import ctypes
from attacking_federate_learning_amd.engine import _vp, _check
def one():
    _check(eng.lib.byz_krum_dev(eng.ctx, _vp(buf.ptr), 100, d, d, 100, 24, 0, _vp(out.ptr), None, None))
for _ in range(20):
    one()
eng.synchronize()
t0 = time.perf_counter()
for _ in range(rounds):
    one()
t1 = time.perf_counter()
eng.synchronize()
t2 = time.perf_counter()
print('D=%d: %.1f us per round to enqueue, %.1f us per round until the GPU is done (%d rounds, raw C ABI calls, no allocation)' % (
    d, (t1 - t0) / rounds * 1e6, (t2 - t0) / rounds * 1e6, rounds))
eng.check()
