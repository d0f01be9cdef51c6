"""A wider seeded sweep of the Bulyan / Krum selection against the C oracle than the test-suite's fixed cases (a one-off
confidence check after a change to the loop; torch-free): for every (N, seed, family) the engine's selection on a distance
matrix must be the reference's pick for pick (oracle/scale.py, the reference's own arithmetic).

    python scripts/selection_sweep.py [seeds]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from attacking_federate_learning_amd.engine import Engine   # noqa: E402
from oracle import scale   # noqa: E402


def distances(seed, n, dim, identical):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    pts *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if identical:
        pts[:identical] = pts[:identical].mean(axis=0)
    p64 = pts.astype(np.float64)
    sq = (p64 * p64).sum(1)
    d = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (p64 @ p64.T), 0.0)).astype(np.float32)
    d = np.minimum(d, d.T)
    if identical:
        d[:identical, :identical] = 0.0
        d[:identical, :] = d[0, :]
        d[:, :identical] = d[:, [0]]
    np.fill_diagonal(d, np.inf)
    return d


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    eng = Engine(0)
    bad = 0
    for n in (300, 777, 1500, 2600, 4000):
        f = int(n * 0.24)
        for seed in range(seeds):
            for dim, identical in ((16, 0), (2000, 0), (16, f), (300, f // 2)):
                d = distances(7000 + 13 * seed + n, n, dim, identical)
                got = np.asarray(eng.bulyan_select(d, n, f)).tolist()
                want = scale.bulyan_selection(d, n, f)
                krum_ok = eng.krum_select(d, n, f) == scale.krum_pick(d, n, f)
                ok = got == want and krum_ok
                bad += 0 if ok else 1
                print('N=%4d seed=%d dim=%4d identical=%4d: %s (%d rows re-scored)' % (
                    n, seed, dim, identical, 'ok' if ok else 'MISMATCH at pick %d' % next(
                        (i for i, (a, b) in enumerate(zip(got, want)) if a != b), -1), eng.bulyan_rescored()), flush=True)
    eng.check()
    print('mismatches: %d' % bad)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
