"""Operand split of the long-K Gram: the one-pass kernel (sampled scale + exact redo) against round 2's two-pass kernel
(torch-free GPU probe): per-kernel times and how far the two Grams are apart."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd.engine import Engine   # noqa: E402


def main():
    eng = Engine(0)
    n, d = 4000, int(sys.argv[1]) if len(sys.argv) > 1 else 262144 + 64
    rng = np.random.default_rng(2)
    g = rng.standard_normal((n, d), dtype=np.float32)
    g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    g[17, 5000:5008] *= 1e3          # an outlier 1000 x the row's scale: that block must be redone exactly
    g[33, :20000] = 0.0              # sampled columns all zero in the first chunks
    buf = eng.to_device(g)
    out = {}
    for two_pass in ('1', '0'):
        os.environ['BYZ_GRAM_SPLIT_TWO_PASS'] = two_pass
        eng.gram(buf)
        eng.check()
        eng.timing(True)
        for _ in range(3):
            res = eng.gram(buf)
        eng.check()
        t = eng.timing_read()
        eng.timing(False)
        out[two_pass] = res.numpy()
        print('two_pass=%s: plane_split %.3f ms, gram_tile %.3f ms per call (%.2f TB/s over read + written bytes)' % (
            two_pass, t['plane_split']['total_ms'] / 3, t['gram_tile']['total_ms'] / 3,
            (4.0 * n * d + 4.0 * 4096 * d) / (t['plane_split']['total_ms'] / 3 * 1e-3) / 1e12), flush=True)
    a, b = out['1'], out['0']
    norms = np.sqrt(np.diag(a))
    rel = np.abs(a - b) / (norms[:, None] * norms[None, :])
    print('max |gram(one pass) - gram(two pass)| / (|gi||gj|) = %.3e; bitwise equal entries: %.4f%%' % (
        rel.max(), 100.0 * np.mean(a == b)), flush=True)


if __name__ == '__main__':
    main()
