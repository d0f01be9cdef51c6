#!/usr/bin/env python3
"""A golden beyond 16,384 clients, minted by the UNMODIFIED reference here: Krum at N = 16,400 (csrc/large_rows.hip).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_beyond.py          # ~25 min, ~25 GB: imports /root/reference, runs it
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_beyond.py --search # how the seed was chosen (fp64 margin of the decision)

The reference's `_krum_create_distances` (defences.py:16-21) is 1.3e8 `np.linalg.norm` calls into a dict of dicts at this size and its
`krum` (defences.py:23-42) sorts 16,400 lists of 16,399 values; Bulyan (8,528 dependent picks, each of them that) is out of reach of the
reference itself -- days -- and stays with the C oracle (tests/test_gpu_large_rows.py).  D = 16 keeps a norm at ~4 us.

Stored (tests/golden/beyond_sizes.npz): the checksum of the regenerated input, the reference's Krum index, SAMPLED ROWS of its distance
dict, the fp64 margin of the decision.  Reads /root/reference/defences.py (imported, never copied).
"""
import os
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import baseline_inputs  # noqa: E402

CASE = dict(name='B0_krum_16400x16', kind='krum', n=16400, d=16, f=3936, seed=7001)
SAMPLED = [0, 1, 2, 8199, 16398, 16399]
MIN_MARGIN = 1e-5       # five times tau = 16 eps_32


def margin_of(seed):
    from oracle import ideal, scale
    g = baseline_inputs.make(dict(CASE, seed=seed))
    dist = ideal.distance_matrix(g).astype(np.float32)
    idx, margin, _ = scale.krum_pick(dist, CASE['n'], CASE['f'], mode='ideal', with_scores=True)
    return idx, float(margin)


def main():
    if '--search' in sys.argv:
        for seed in range(CASE['seed'], CASE['seed'] + 40):
            idx, margin = margin_of(seed)
            print('seed %d: index %d, fp64 margin %.3e' % (seed, idx, margin), flush=True)
            if margin >= MIN_MARGIN:
                print('  -> seed %d' % seed)
                break
        return
    sys.path.insert(0, '/root/reference')
    import defences as ref_defences
    g = baseline_inputs.make(CASE)
    n, f = CASE['n'], CASE['f']
    t0 = time.perf_counter()
    distances = ref_defences._krum_create_distances(g)
    print('reference distances: %.0f s' % (time.perf_counter() - t0), flush=True)
    rows = np.full((len(SAMPLED), n), np.inf, dtype=np.float32)
    for k, u in enumerate(SAMPLED):
        for j, v in distances[u].items():
            rows[k, j] = v
    t0 = time.perf_counter()
    index = ref_defences.krum(g, n, f, distances=distances, return_index=True)
    print('reference krum: index %d, %.0f s' % (index, time.perf_counter() - t0), flush=True)
    _, margin = margin_of(CASE['seed'])
    np.savez_compressed(os.path.join(HERE, 'beyond_sizes.npz'), checksum=baseline_inputs.checksum(g), krum_index=np.int64(index),
                        sampled_rows=np.asarray(SAMPLED, dtype=np.int64), distance_rows=rows, krum_margin=np.float64(margin))
    print('wrote beyond_sizes.npz (margin %.3e)' % margin)


if __name__ == '__main__':
    main()
