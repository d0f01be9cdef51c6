"""PCIe-inclusive timing of the numpy drop-in path (what main.py sees): host matrix in, host vector out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from attacking_federate_learning_amd import defences
rng = np.random.default_rng(0)
for name, n, d, f in (('Krum', 100, 79510, 24), ('Krum', 100, 21840, 24), ('TrimmedMean', 100, 79510, 24), ('Bulyan', 100, 79510, 24), ('NoDefense', 100, 79510, 24)):
    g = rng.standard_normal((n, d)).astype(np.float32)
    fn = defences.defend[name]
    for _ in range(3): fn(g, n, f)
    t0 = time.perf_counter()
    k = 20
    for _ in range(k): fn(g, n, f)
    dt = (time.perf_counter() - t0) / k
    print('%-12s N=%d D=%d host numpy in -> numpy out: %.3f ms per call (%.1f MB over PCIe)' % (name, n, d, dt * 1e3, g.nbytes / 1e6), flush=True)
