#!/usr/bin/env python3
"""Golden vectors at BASELINE.json's OWN small configurations, minted by the UNMODIFIED reference in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_baseline.py

`make_golden.py` stops at 40 x 500 / 33 x 1000 matrices because it stores its inputs.  The configurations the reference
itself runs (configs[0]: N = 10 clients of MnistNet, D = 79,510, Krum; configs[1]: N = 100, D = 79,510 and 21,840;
configs[2]: N = 1000, trimmed mean) cost the reference milliseconds to seconds, so they are minted here with SEEDED
inputs (`tests/golden/baseline_inputs.py` regenerates them bit for bit on any box: numpy's PCG64 streams) and only the
reference's OUTPUTS are stored -- index, selection, the dense distance matrix, sampled columns of the aggregate -- a few
hundred KiB in all.  A checksum of every regenerated input is stored as well, so a box whose numpy generates another
stream fails on the checksum, not on a parity assertion.

Reads /root/reference/{defences,malicious}.py (imported, never copied).  Writes tests/golden/baseline_sizes.npz.
"""
import os
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, '/root/reference')
import defences as ref_defences  # noqa: E402
import malicious as ref_malicious  # noqa: E402

import baseline_inputs as inputs  # noqa: E402


class FakeUser:
    def __init__(self, grads):
        self.grads = grads
        self.original_params = None
        self.learning_rate = None


def reference_attack(g, m, z):
    """Rows 0..m-1 := the reference's own drift vector (malicious.py:10-27); returns the vector."""
    users = [FakeUser(g[i].copy()) for i in range(m)]
    ref_malicious.DriftAttack(z).attack(users)
    for i in range(m):
        g[i] = users[i].grads
    return users[0].grads


def dense(dist_dict, n):
    out = np.full((n, n), np.inf, dtype=np.float32)
    for i, row in dist_dict.items():
        for j, v in row.items():
            out[i, j] = v
    return out


def bulyan_selection(g, n, f, dist):
    """defences.py:59-68 replayed with the reference's own krum() on a COPY of its distance dict: the picks."""
    dist = {i: dict(row) for i, row in dist.items()}
    picks = []
    while len(picks) < n - 2 * f:
        idx = ref_defences.krum(g, n - len(picks), f, dist, True)
        picks.append(idx)
        dist.pop(idx)
        for r in dist:
            dist[r].pop(idx)
    return np.asarray(picks, dtype=np.int64)


def main():
    out = {}

    def put(case, **kv):
        for k, v in kv.items():
            out['%s/%s' % (case, k)] = np.asarray(v)

    for case in inputs.CASES:
        t0 = time.perf_counter()
        name, kind = case['name'], case['kind']
        g = inputs.make(case)
        n, d = g.shape
        put(name, checksum=inputs.checksum(g))
        if case.get('attack'):
            # the GPU test applies ITS attack to the seeded rows; the reference's vector is the pin for that step too
            drift = reference_attack(g, case['attack'], case.get('z', 1.5))
            put(name, drift_cols=inputs.sample_columns(case, d), drift=drift[inputs.sample_columns(case, d)])
        if kind in ('krum', 'krum+bulyan'):
            f = case['f']
            dist = ref_defences._krum_create_distances(g)
            put(name, dist=dense(dist, n), index=ref_defences.krum(g, n, f, dist, True))
            if kind == 'krum+bulyan':
                cols = inputs.sample_columns(case, d)
                put(name, selection=bulyan_selection(g, n, f, dist), out_cols=cols,
                    out=ref_defences.bulyan(g, n, f)[cols])
        elif kind == 'trimmed_mean':
            put(name, out=ref_defences.trimmed_mean(g, n, case['c']))
        print('%-34s %6.1f s' % (name, time.perf_counter() - t0), flush=True)

    path = os.path.join(HERE, 'baseline_sizes.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KiB' % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
