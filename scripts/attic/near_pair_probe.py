import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from attacking_federate_learning_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(5)
n, d = 3000, 20000
g = rng.standard_normal((n, d), dtype=np.float32)
g[:700] = g[0]
buf = eng.to_device(g)
dist = eng.pairwise_distances(buf)
print('near pairs listed:', eng.near_pairs_count())
dd = dist.numpy()
print('zeros among twins:', bool((dd[:700, :700][~np.eye(700, dtype=bool)] == 0).all()), 'rows identical:', bool((dd[5, 700:] == dd[0, 700:]).all()))
eng.timing(True)
dist = eng.pairwise_distances(buf)
t = eng.timing_read(); eng.timing(False)
print({k: round(v['total_ms'], 3) for k, v in t.items()})
