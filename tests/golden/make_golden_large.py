#!/usr/bin/env python3
"""Golden vectors at sizes that drive the engine's production large-N kernels, minted by the UNMODIFIED reference here.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_large.py            # ~5 min: imports /root/reference, runs it
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_large.py --search   # how the seeds in large_inputs.py were chosen

What is stored per case (tests/golden/large_sizes.npz): the checksum of the regenerated input, the reference's drift vector at
sampled columns (malicious.py:10-36), SAMPLED ROWS of the reference's distance dict (defences.py:16-21), its Krum index
(defences.py:23-42), its Bulyan selection and sampled columns of its aggregate (defences.py:55-70), and the fp64 margins of
the decisions (oracle.ideal / oracle.scale on the same matrix) that the seed search looked at.

Reads /root/reference/{defences,malicious}.py (imported, never copied).
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

import large_inputs as inputs  # noqa: E402

MIN_MARGIN = 1e-5       # five times tau = 16 eps_32


def oracle_attack(g, m, z):
    from oracle import faithful
    g[:m] = faithful.drift_vector(g[:m].copy(), z)


def margins_of(case, seed):
    """fp64 margins of every decision the reference will make on the seeded matrix (the oracle restates it; pinned elsewhere)."""
    from oracle import ideal, scale
    c = dict(case, seed=seed)
    g = inputs.make(c)
    if c.get('attack'):
        oracle_attack(g, c['attack'], c.get('z', 1.5))
    dist = ideal.distance_matrix(g).astype(np.float32)
    n, f = c['n'], c['f']
    _, krum_margin, _ = scale.krum_pick(dist, n, f, mode='ideal', with_scores=True)
    out = {'krum': float(krum_margin)}
    if c['kind'] == 'krum+bulyan':
        _, m = scale.bulyan_selection(dist, n, f, mode='ideal', with_margins=True)
        # under the attack the identical rows tie EXACTLY (margin 0) and the visit order decides: those picks are not noise
        out['bulyan_min_nonzero'] = float(m[m > 0].min()) if np.any(m > 0) else float('inf')
        out['bulyan_exact_ties'] = int((m == 0).sum())
        out['bulyan_margins'] = m
    return out


def search():
    for case in inputs.CASES:
        for seed in range(case['seed'], case['seed'] + 60):
            t0 = time.perf_counter()
            m = margins_of(case, seed)
            worst = min(m['krum'] if m['krum'] > 0 else float('inf'), m.get('bulyan_min_nonzero', float('inf')))
            print('%-32s seed %d: krum margin %.3e, smallest non-zero bulyan margin %.3e (%d exact ties)  %.1f s'
                  % (case['name'], seed, m['krum'], m.get('bulyan_min_nonzero', float('nan')), m.get('bulyan_exact_ties', 0),
                     time.perf_counter() - t0), flush=True)
            if worst >= MIN_MARGIN:
                print('  -> seed %d' % seed)
                break


class FakeUser:
    def __init__(self, grads):
        self.grads = grads
        self.original_params = None
        self.learning_rate = None


def main():
    sys.path.insert(0, '/root/reference')
    import defences as ref_defences
    import malicious as ref_malicious
    out = {}

    def put(case, **kv):
        for k, v in kv.items():
            out['%s/%s' % (case, k)] = np.asarray(v)

    for case in inputs.CASES:
        t0 = time.perf_counter()
        name = case['name']
        g = inputs.make(case)
        n, d = g.shape
        f = case['f']
        put(name, checksum=inputs.checksum(g))
        cols = inputs.sample_columns(case, d)
        if case.get('attack'):
            users = [FakeUser(g[i].copy()) for i in range(case['attack'])]
            ref_malicious.DriftAttack(case.get('z', 1.5)).attack(users)           # malicious.py:10-36, unmodified
            for i in range(case['attack']):
                g[i] = users[i].grads
            put(name, drift_cols=cols, drift=users[0].grads[cols])
        dist = ref_defences._krum_create_distances(g)                             # defences.py:16-21, unmodified
        rows = inputs.sample_rows(case)
        sampled = np.full((len(rows), n), np.inf, dtype=np.float32)
        for k, r in enumerate(rows):
            for j, v in dist[int(r)].items():
                sampled[k, j] = v
        put(name, dist_rows=rows, dist=sampled, index=ref_defences.krum(g, n, f, dist, True))     # defences.py:23-42
        m = margins_of(case, case['seed'])
        put(name, krum_margin=m['krum'])
        print('%-32s distances + krum %6.1f s (index %d, fp64 margin %.2e)' % (name, time.perf_counter() - t0,
                                                                              int(out[name + '/index']), m['krum']), flush=True)
        if case['kind'] == 'krum+bulyan':
            # defences.py:59-68 with the reference's own krum() on a COPY of its distance dict: the picks, in order
            work = {i: dict(row) for i, row in dist.items()}
            picks = []
            while len(picks) < n - 2 * f:
                idx = ref_defences.krum(g, n - len(picks), f, work, True)
                picks.append(idx)
                work.pop(idx)
                for r in work:
                    work[r].pop(idx)
            agg = ref_defences.bulyan(g, n, f)                                    # defences.py:55-70, unmodified
            put(name, selection=np.asarray(picks, dtype=np.int64), out_cols=cols, out=agg[cols],
                bulyan_margins=m['bulyan_margins'])
            print('%-32s bulyan           %6.1f s (smallest non-zero fp64 margin %.2e, %d exact ties)'
                  % (name, time.perf_counter() - t0, m['bulyan_min_nonzero'], m['bulyan_exact_ties']), flush=True)
    path = os.path.join(HERE, 'large_sizes.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KiB' % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    if '--search' in sys.argv:
        search()
    else:
        main()
