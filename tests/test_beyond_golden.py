"""The UNMODIFIED reference's Krum at N = 16,400 clients -- beyond the 16,384 rows the LDS-resident selection kernels hold, where
csrc/large_rows.hip takes over (tests/golden/beyond_sizes.npz, minted by tests/golden/make_golden_beyond.py from the imported
reference: ~25 minutes and ~25 GB of dict there; the input is regenerated from its seed, only outputs are stored).

Without it the kernels for more than 16,384 rows would be tied to the reference only through the oracle chain
(tests/test_gpu_large_rows.py: oracle/scale.py -> oracle/faithful.py -> the reference at small N).  Bulyan at this size is out of the
reference's own reach (8,528 dependent picks, each one 16,400 sorts of 16,399 values) and stays with the C oracle.

Bars: the reference's index EXACTLY (the seed was chosen for an fp64 margin of 5.7e-4, 300 times tau = 16 eps), its distances to 1e-5.
CPU half: the oracle reproduces the reference at this size.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import baseline_inputs  # noqa: E402

from oracle import ideal, scale  # noqa: E402

CASE = dict(name='B0_krum_16400x16', kind='krum', n=16400, d=16, f=3936, seed=7001)       # = make_golden_beyond.py's


@pytest.fixture(scope='module')
def beyond():
    z = np.load(os.path.join(HERE, 'golden', 'beyond_sizes.npz'))
    g = baseline_inputs.make(CASE)
    assert np.array_equal(baseline_inputs.checksum(g), z['checksum']), \
        'this box regenerates another input stream than the one the golden outputs were minted on'
    return g, {k: z[k] for k in z.files}


def rows_close(got_rows, want_rows, rows, rtol):
    ok = True
    for k, r in enumerate(rows):
        got, want = np.asarray(got_rows[k], dtype=np.float64), np.asarray(want_rows[k], dtype=np.float64)
        live = np.ones(len(want), dtype=bool)
        live[int(r)] = False
        ok &= bool(np.isinf(want[int(r)]))                      # the dict has no self entry (defences.py:18-20)
        ok &= bool(np.allclose(got[live], want[live], rtol=rtol, atol=0.0))
    return ok


def fp64_distances(g, rows=1024):
    """oracle.ideal.distance_matrix's arithmetic (the Gram identity in fp64, rounded to fp32 once) a block of rows at a time: its
    three n x n fp64 temporaries and the transposed pass take two minutes at this size on the build container.  (No identical rows in
    this case, so the symmetrising minimum of the original is not needed.)"""
    p = g.astype(np.float64)
    n = len(p)
    sq = (p * p).sum(1)
    out = np.empty((n, n), dtype=np.float32)
    for s in range(0, n, rows):
        d2 = sq[s:s + rows, None] + sq[None, :] - 2.0 * (p[s:s + rows] @ p.T)
        np.maximum(d2, 0.0, out=d2)
        np.sqrt(d2, out=d2)
        out[s:s + rows] = d2
    np.fill_diagonal(out, np.inf)
    return out


def test_oracle_reproduces_the_reference_beyond_16384_rows(beyond):
    g, want = beyond
    assert float(want['krum_margin']) > 1e-5
    small = ideal.distance_matrix(g[:300]).astype(np.float32)
    dist = fp64_distances(g)
    assert np.allclose(dist[:300, :300], small, rtol=1e-6, atol=0.0, equal_nan=True)      # the same arithmetic as the oracle's
    assert rows_close(dist[want['sampled_rows']], want['distance_rows'], want['sampled_rows'], 1e-5)
    assert scale.krum_pick(dist, CASE['n'], CASE['f']) == int(want['krum_index'])


@pytest.mark.gpu
def test_gpu_krum_is_the_references_beyond_16384_rows(eng, beyond):
    """defences.py:16-42 at N = 16,400: the engine's distances against the reference's dict rows, its index from its own distances
    (row sort, scores and argmin of csrc/large_rows.hip), and the row `krum` returns."""
    torch = pytest.importorskip('torch')
    g, want = beyond
    n, f = CASE['n'], CASE['f']
    gd = torch.from_numpy(g).cuda()
    dist = eng.pairwise_distances(gd).numpy()
    assert rows_close(dist[want['sampled_rows']], want['distance_rows'], want['sampled_rows'], 1e-5)
    assert eng.krum(gd, n, f, return_index=True) == int(want['krum_index'])
    assert eng.krum_select(dist, n, f) == int(want['krum_index'])
    assert torch.equal(eng.krum(gd, n, f), gd[int(want['krum_index'])])
