"""Drift attack with write-back (malicious.py:10-36) on m rows x D columns (torch-free GPU probe): the statistics pass and the
broadcast of the attack vector into the m rows, per-kernel times."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Engine
eng = Engine(0)
m, d = 2400, int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
g = np.random.default_rng(3).standard_normal((m, d), dtype=np.float32)
buf = eng.to_device(g)
eng.drift_attack(buf, 1.5, write_back=True)
eng.timing(True)
for _ in range(5):
    drift = eng.drift_attack(buf, 1.5, write_back=True)[0]
t = eng.timing_read(); eng.timing(False)
print({k: round(v['total_ms'] / 5, 3) for k, v in t.items()}, flush=True)
rows = buf.numpy()
vec = np.asarray(drift.numpy() if hasattr(drift, 'numpy') else drift)
print('rows all equal to the returned vector:', bool((rows == vec[None, :]).all()), '| %.2f TB/s written' % (4.0 * m * d / (t['misc']['total_ms'] / 5 * 1e-3) / 1e12))
