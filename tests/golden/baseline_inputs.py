"""Seeded inputs of tests/golden/baseline_sizes.npz (the reference's outputs at BASELINE.json's own small configurations).

The inputs are NOT stored: `make(case)` regenerates them bit for bit (numpy PCG64 streams; `checksum` is stored beside the
outputs and a box that generates another stream fails on it).  Families as in SURVEY.md 8(d): `scaled` = row i of N(0, 1)
times 1 + 0.5 pi(i) / N -- well-separated Krum scores, so that the reference's decisions do not hang on the last bits of
its OpenBLAS sdot (index parity is then a statement about the algorithm, not about summation order).
"""
import numpy as np

CASES = [
    # BASELINE configs[0]: main.py's own case -- 10 clients of MnistNet (D = 79,510), Krum, mal_prop 0.24 -> f = 2
    dict(name='c1_krum_10x79510', kind='krum', n=10, d=79510, f=2, seed=4101),
    # configs[1]: N = 100, D = 79,510 (MnistNet) and 21,840 (the survey's CNN), f = 24; Bulyan on the same distances
    dict(name='c2_100x79510', kind='krum+bulyan', n=100, d=79510, f=24, seed=4102),
    dict(name='c2_100x21840', kind='krum+bulyan', n=100, d=21840, f=24, seed=4103),
    # the same with the attack's 24 identical rows (malicious.py:26-27): exact ties, resolved by visit order
    dict(name='c2_attacked_100x79510', kind='krum+bulyan', n=100, d=79510, f=24, seed=4104, attack=24, z=1.5),
    dict(name='c2_attacked_100x21840', kind='krum+bulyan', n=100, d=21840, f=24, seed=4105, attack=24, z=1.5),
    # configs[2]: N = 1000, trim 200, on 512 seeded columns (the reference needs 0.5 ms per column)
    dict(name='c3_tm_1000x512', kind='trimmed_mean', n=1000, d=512, c=200, seed=4106),
    dict(name='c3_tm_attacked_1000x256', kind='trimmed_mean', n=1000, d=256, c=200, seed=4107, attack=240, z=1.5),
]

SAMPLED_COLUMNS = 4096


def make(case):
    """The seeded N x D fp32 matrix of a case, BEFORE the attack (the attack is applied by whoever is under test)."""
    rng = np.random.default_rng(case['seed'])
    n, d = case['n'], case['d']
    g = rng.standard_normal((n, d)).astype(np.float32)
    s = (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)
    return g * s[:, None]


def checksum(g):
    """A few numbers that pin the generated matrix: its fp64 sum, its fp64 sum of squares, three sampled entries."""
    g64 = g.astype(np.float64)
    n, d = g.shape
    return np.array([g64.sum(), (g64 * g64).sum(), g64[0, 0], g64[n // 2, d // 3], g64[n - 1, d - 1]])


def sample_columns(case, d):
    """The columns at which a D-vector of the case is stored (all of them when D is small)."""
    if d <= SAMPLED_COLUMNS:
        return np.arange(d, dtype=np.int64)
    rng = np.random.default_rng(case['seed'] + 100000)
    return np.sort(rng.choice(d, size=SAMPLED_COLUMNS, replace=False)).astype(np.int64)
