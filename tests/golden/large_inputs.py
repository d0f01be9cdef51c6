"""Seeded inputs of tests/golden/large_sizes.npz: the UNMODIFIED reference's outputs at sizes that send the engine through its
PRODUCTION large-N kernels (missing 4: every reference-minted Krum / Bulyan golden had N <= 100 and therefore ran
krum_small.hip only).

    case                       engine path it pins to the reference
    L0 krum 2900 x 17,000      gram_planes.hip f16x2 at its smallest shape (no identical rows) + the grid row sort
    L1 krum 3000 x 24,640      dedup.hip (720 identical rows) + gram_planes.hip f16x2 (N >= 2817, D > 16,384) + the grid row sort
    L2 bulyan 600 x 20,000     dedup.hip (144 identical rows) + gram.hip bf16x3 tiles + bulyan_grid_kernel on 3 workgroups + the
                               ring selection of the second stage (312 rows)
    L3 bulyan 300 x 12,000     the same without the attack: no identical rows, 2 workgroups

Inputs are regenerated bit for bit from the seed (tests/golden/baseline_inputs.py::make; a checksum is stored); only the
reference's OUTPUTS are stored.  The seeds were CHOSEN (make_golden_large.py --search) so that every decision the reference
makes on these matrices has an fp64 margin above 1e-5 -- five times tau = 16 eps, fifty times the noise of the engine's
distances against the reference's sdot -- or is an EXACT tie between the attack's identical rows (margin 0, decided by the visit
order 1, 0, 2, ...): index equality with the reference is then a statement about the algorithm and not about the last bits of
its OpenBLAS sdot.  The margins are stored beside the outputs.  (Of 440 seeds tried for L2 the best smallest margin was 1.4e-5:
312 sequential picks among 600 rows nearly always pass through one closer than that.)
"""
import numpy as np

import baseline_inputs

CASES = [
    dict(name='L0_krum_2900x17000', kind='krum', n=2900, d=17000, f=696, seed=6001),
    dict(name='L1_krum_attacked_3000x24640', kind='krum', n=3000, d=24640, f=720, attack=720, z=1.5, seed=6101),
    dict(name='L2_bulyan_attacked_600x20000', kind='krum+bulyan', n=600, d=20000, f=144, attack=144, z=1.5, seed=6618),
    dict(name='L3_bulyan_300x12000', kind='krum+bulyan', n=300, d=12000, f=72, seed=6351),
]

SAMPLED_ROWS = 24


def make(case):
    return baseline_inputs.make(case)


checksum = baseline_inputs.checksum
sample_columns = baseline_inputs.sample_columns


def sample_rows(case):
    """The rows of the reference's distance matrix that are stored: the first and last malicious rows, their honest
    neighbours, the last rows of the matrix (the ragged last tile) and a seeded handful in between."""
    n, m = case['n'], case.get('attack', 0)
    fixed = [0, 1, max(m - 1, 0), m, min(m + 1, n - 1), n // 2, n - 2, n - 1]
    rng = np.random.default_rng(case['seed'] + 200000)
    rest = rng.choice(n, size=SAMPLED_ROWS, replace=False).tolist()
    rows = []
    for r in fixed + rest:
        if r not in rows:
            rows.append(int(r))
    return np.asarray(rows[:SAMPLED_ROWS], dtype=np.int64)
