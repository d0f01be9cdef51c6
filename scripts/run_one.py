"""Run one kernel family a few times (profiling target).  usage: run_one.py tm N D [c] | gram N D | attack M D | bulyan N D"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from attacking_federate_learning_amd.engine import get_engine
eng = get_engine()
what, n, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(os.environ.get('ITERS', '5'))
gen = torch.Generator(device='cuda').manual_seed(0)
g = torch.randn((n, d), device='cuda', generator=gen)
for _ in range(iters):
    if what == 'tm':
        eng.trimmed_mean(g, n, int(sys.argv[4]) if len(sys.argv) > 4 else n // 5)
    elif what == 'gram':
        eng.gram(g)
    elif what == 'attack':
        eng.drift_attack(g, 1.5)
    elif what == 'bulyan':
        eng.bulyan(g, n, int(n * 0.24))
    elif what == 'krum':
        eng.krum(g, n, int(n * 0.24), return_index=True)
torch.cuda.synchronize()
print('done', what, n, d)
