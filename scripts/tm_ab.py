"""Same-box A/B of trimmed-mean variants selected by environment variables (torch-free): alternates the settings several
times on one matrix per shape, because two gpurun visits land on boxes that differ by up to 10%."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd.engine import Engine
eng = Engine(0)
var = sys.argv[1]
values = sys.argv[2].split(',')
scale = int(os.environ.get('TM_AB_SCALE', '1'))   # (wider matrices: more tiles per launch)
shapes = [(1000, scale << 18, 200), (2080, scale << 17, 1920), (5200, scale << 16, 4800)]
for rows, cols, corrupted in shapes:
    buf = eng.to_device(np.random.default_rng(rows).standard_normal((rows, cols), dtype=np.float32))
    res = {v: [] for v in values}
    for rep in range(4):
        for v in values:
            os.environ[var] = v
            eng.trimmed_mean(buf, rows, corrupted); eng.synchronize()
            t0 = time.perf_counter()
            outs = [eng.trimmed_mean(buf, rows, corrupted) for _ in range(20)]
            eng.synchronize()
            res[v].append((time.perf_counter() - t0) / 20 * 1e3)
            del outs
    print('%5d x %7d: ' % (rows, cols) + '   '.join('%s=%s: %s ms' % (var, v, ' '.join('%.3f' % t for t in res[v])) for v in values), flush=True)
