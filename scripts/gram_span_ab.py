"""Same-box A/B of the deferred Gram tile kernel by span (round 6): chunks a workgroup walks without leaving (BYZ_GRAM_KSPAN),
at one launch of configs[3] (N = 4000, 1,000,448 columns) or of configs[4]'s slice (N = 10,000, 401,408 columns), on the device
(torch generates the matrix).  Alternates the settings, prints the tile kernel's and the reduce kernel's time per call and
whether the Gram is bitwise the first setting's.

    python scripts/gram_span_ab.py 4000 1000448 BYZ_GRAM_KSPAN=1 BYZ_GRAM_KSPAN=4 BYZ_GRAM_KSPAN=4,BYZ_GRAM_ROUND=0
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
from attacking_federate_learning_amd.engine import get_engine   # noqa: E402


def main():
    n, d = int(sys.argv[1]), int(sys.argv[2])
    settings = sys.argv[3:] or ['BYZ_GRAM_KSPAN=1', 'BYZ_GRAM_KSPAN=4']
    reps = int(os.environ.get('REPS', '3'))
    calls = int(os.environ.get('CALLS', '3'))
    eng = get_engine()
    gen = torch.Generator(device='cuda').manual_seed(n)
    g = torch.randn((n, d), device='cuda', generator=gen)
    g *= (1.0 + 0.5 * torch.rand((n, 1), device='cuda', generator=gen))
    touched = set()
    ref = None
    for rep in range(reps):
        for setting in settings:
            for key in touched:
                os.environ.pop(key, None)
            for kv in setting.split(','):
                key, val = kv.split('=', 1)
                os.environ[key] = val
                touched.add(key)
            res = eng.gram(g)            # warm (workspaces, tile order)
            eng.check()
            eng.timing(True)
            for _ in range(calls):
                res = eng.gram(g)
            eng.check()
            t = eng.timing_read()
            eng.timing(False)
            if ref is None:
                ref = res.clone()
            tile = t['gram_tile']['total_ms'] / calls
            red = t.get('gram_reduce', {'total_ms': 0.0})['total_ms'] / calls
            print('rep %d  %-44s gram_tile %8.3f ms  gram_reduce %6.3f ms  plane_split %7.3f ms  (%.1f TF-eq, %.4f of the f16x2 roof)  bitwise: %s'
                  % (rep, setting, tile, red, t['plane_split']['total_ms'] / calls, 1.0 * n * n * d / (tile * 1e-3) / 1e12,
                     1.0 * n * n * d / (tile * 1e-3) / (2.5e15 / 3), bool(torch.equal(res, ref))), flush=True)


if __name__ == '__main__':
    main()
