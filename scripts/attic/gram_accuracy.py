"""Accuracy of the long-K Gram (fp16 x 2 planes) against fp64 on sampled rows, for several K (torch-free)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd.engine import Engine   # noqa: E402


def main():
    eng = Engine(0)
    n = 3000
    rng = np.random.default_rng(1)
    sample = np.concatenate([np.arange(0, n, 97), [n - 1, n - 2, 2468 % n]])
    for d in [int(x) for x in (sys.argv[1:] or [16384 + 64, 16400, 16384 + 32, 16384 + 48, 24640, 24576 + 16])]:
        g = rng.standard_normal((n, d), dtype=np.float32)
        g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
        buf = eng.to_device(g)
        got = eng.gram(buf).numpy()
        eng.check()
        g64 = g.astype(np.float64)
        want = g64[sample] @ g64.T
        norms = np.sqrt((g64 * g64).sum(1))
        err = np.abs(got[sample] - want) / (norms[sample][:, None] * norms[None, :])
        k = np.unravel_index(err.argmax(), err.shape)
        diag = np.abs(got[sample, sample] - want[np.arange(len(sample)), sample]) / norms[sample] ** 2
        print('D=%6d  max |c - c64| / (|gi||gj|) = %.2e at (%d, %d);  diagonal %.2e;  last row max %.2e' % (
            d, err.max(), sample[k[0]], k[1], diag.max(), err[-3].max()), flush=True)


if __name__ == '__main__':
    main()
