"""Same-box A/B of the long-K Gram tile kernel (torch-free GPU probe): time per call of the tile kernel under each setting,
alternated, and whether the Gram is bitwise the first setting's.

    python scripts/gram_ab.py BYZ_GRAM_BLOCK_SKIP=0,BYZ_GRAM_BLOCK_SKIP=1 4000 262224
    python scripts/gram_ab.py 0,20 ...          a bare number is a BYZ_GRAM_PLANES_VARIANT (needs a -DBYZ_GRAM_DEBUG_VARIANTS build)
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd.engine import Engine   # noqa: E402


def main():
    variants = sys.argv[1].split(',') if len(sys.argv) > 1 else ['0', '2']
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 262144 + 80
    kind = sys.argv[4] if len(sys.argv) > 4 else 'normal'   # zeros / ones: how much of the time is the DATA (switching power)
    eng = Engine(0)
    rng = np.random.default_rng(2)
    if kind == 'normal':
        g = rng.standard_normal((n, d), dtype=np.float32)
        g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    elif kind == 'rowconst':     # distinct rows (no identical-row shortcut), nothing changes along K
        g = np.repeat((1.0 + np.arange(n, dtype=np.float32) / n)[:, None], d, axis=1)
    elif kind == 'signs':        # +-1: the low plane is zero, one bit per element toggles
        g = (rng.integers(0, 2, (n, d), dtype=np.int8) * 2 - 1).astype(np.float32)
    else:
        g = np.full((n, d), 0.0 if kind == 'zeros' else 1.0, dtype=np.float32)
    print('data: %s' % kind, flush=True)
    buf = eng.to_device(g)
    del g
    ref = None
    for rep in range(2):
        for v in variants:
            opts = v.split(':')                                           # "0:r0" = variant 0 without the round gate
            if '=' in opts[0]:
                key, val = opts[0].split('=', 1)
                os.environ[key] = val
            else:
                os.environ['BYZ_GRAM_PLANES_VARIANT'] = opts[0]
            os.environ['BYZ_GRAM_ROUND'] = '0' if 'r0' in opts[1:] else '32'
            res = eng.gram(buf)
            eng.check()
            eng.timing(True)
            for _ in range(3):
                res = eng.gram(buf)
            eng.check()
            t = eng.timing_read()
            eng.timing(False)
            a = res.numpy()
            if ref is None:
                ref = a
            norms = np.sqrt(np.abs(np.diag(ref))) + 1e-300
            print('variant %s: gram_tile %.3f ms, plane_split %.3f ms per call (%.1f TF-eq); bitwise the first variant: %s (max difference %.2e of |gi||gj|)' % (
                v, t['gram_tile']['total_ms'] / 3, t['plane_split']['total_ms'] / 3,
                1.0 * n * n * d / (t['gram_tile']['total_ms'] / 3 * 1e-3) / 1e12,
                bool(np.array_equal(a, ref)), float(np.max(np.abs(a - ref) / (norms[:, None] * norms[None, :])))), flush=True)


if __name__ == '__main__':
    main()
