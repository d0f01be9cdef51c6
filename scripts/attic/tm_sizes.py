"""trimmed_mean timings across tile geometries (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attacking_federate_learning_amd.engine import get_engine
eng = get_engine()
sizes = [int(a) for a in sys.argv[1:]] or [500, 1000, 1500, 2080, 2500]
d = 1 << 20
gen = torch.Generator(device='cuda').manual_seed(0)
for n in sizes:
    g = torch.randn((n, d), device='cuda', generator=gen)
    for _ in range(2): eng.trimmed_mean(g, n, n // 5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): eng.trimmed_mean(g, n, n // 5)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print('N=%5d D=%d  %8.3f ms  %7.1f GB/s' % (n, d, ms, 4.0 * n * d / ms / 1e6), flush=True)
    del g
