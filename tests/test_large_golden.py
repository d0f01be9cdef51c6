"""The UNMODIFIED reference's outputs at sizes that send the engine through its production large-N kernels
(tests/golden/large_sizes.npz, minted by tests/golden/make_golden_large.py from the imported reference; inputs regenerated from
seeds, only outputs stored).  VERDICT r5, missing 4: until round 6 every reference-minted Krum / Bulyan golden had N <= 100 and
ran krum_small.hip; gram.hip, dedup.hip, gram_planes.hip and the multi-workgroup Bulyan loop were tied to the reference only
through the oracle chain.

    L0  Krum   N = 2900, D = 17,000, f = 696            gram_planes.hip f16x2 (N >= 2817, D > 16,384), grid row sort
    L1  Krum   N = 3000, D = 24,640, f = 720, attacked  + dedup.hip over the attack's 720 identical rows (defences.py:16-42)
    L2  Bulyan N = 600,  D = 20,000, f = 144, attacked  dedup + gram.hip bf16x3 + bulyan_grid_kernel on 3 workgroups + second stage
    L3  Bulyan N = 300,  D = 12,000, f = 72             the same without identical rows (defences.py:55-70)

Bars (north_star): the reference's index and selection EXACTLY, its distances and its aggregate to 1e-5.  The seeds were chosen
so that every decision has an fp64 margin above 1e-5 or is an exact tie of the attack's rows (tests/golden/large_inputs.py).

CPU half: the oracle restatement reproduces what the reference returned at these sizes (so the oracle chain the other large-N
tests hang on is itself pinned here, not only at N <= 100).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import large_inputs as inputs  # noqa: E402

from oracle import faithful, ideal, scale  # noqa: E402

RTOL = ATOL = 1e-5
BY_NAME = {c['name']: c for c in inputs.CASES}
ALL = [c['name'] for c in inputs.CASES]
BULYAN = [c['name'] for c in inputs.CASES if c['kind'] == 'krum+bulyan']


@pytest.fixture(scope='module')
def large():
    z = np.load(os.path.join(HERE, 'golden', 'large_sizes.npz'))
    cases = {}
    for key in z.files:
        case, field = key.split('/', 1)
        cases.setdefault(case, {})[field] = z[key]
    return cases


_made = {}


def seeded(name, large):
    if name not in _made:
        g = inputs.make(BY_NAME[name])
        assert np.array_equal(inputs.checksum(g), large[name]['checksum']), \
            'this box regenerates another input stream than the one the golden outputs were minted on'
        _made.clear()            # one matrix at a time (up to 296 MB)
        _made[name] = g
    return _made[name].copy()


def rows_close(dist_rows, want_rows, rows, rtol=RTOL):
    """Sampled rows of a distance matrix against the reference's dict rows (+inf where the dict has no entry: the diagonal).
    Exact zeros (the attack's identical rows: norm(a - a) == 0.0 in the reference) must be exact zeros."""
    ok = True
    for k, r in enumerate(rows):
        got, want = np.asarray(dist_rows[k], dtype=np.float64), np.asarray(want_rows[k], dtype=np.float64)
        live = np.ones(len(want), dtype=bool)
        live[int(r)] = False
        ok &= bool(np.isinf(want[int(r)]))
        ok &= bool(np.array_equal(got[live] == 0.0, want[live] == 0.0))
        ok &= bool(np.allclose(got[live], want[live], rtol=rtol, atol=0.0))
    return ok


# ---- CPU: the oracle at these sizes ---------------------------------------------------------------------------------
def attacked_by_oracle(name, large):
    case, g = BY_NAME[name], seeded(name, large)
    if case.get('attack'):
        drift = faithful.drift_vector(g[:case['attack']], case['z'])
        assert np.array_equal(drift[large[name]['drift_cols']], large[name]['drift'])      # bit for bit the reference's vector
        g[:case['attack']] = drift
    return g


@pytest.mark.parametrize('name', ALL)
def test_oracle_reproduces_the_reference_at_large_sizes(large, name):
    """fp64 distances rounded to fp32 against the reference's sdot distances; the C selection oracle in the reference's own
    arithmetic (sequential fp32 sums) on those distances returns the reference's index and selection."""
    case, want = BY_NAME[name], large[name]
    n, f = case['n'], case['f']
    g = attacked_by_oracle(name, large)
    dist = ideal.distance_matrix(g).astype(np.float32)
    rows = want['dist_rows']
    assert rows_close(dist[rows], want['dist'], rows)
    assert scale.krum_pick(dist, n, f) == int(want['index'])
    if case['kind'] == 'krum+bulyan':
        assert scale.bulyan_selection(dist, n, f) == want['selection'].tolist()
        sel = want['selection']
        cols = want['out_cols']
        assert np.allclose(faithful.trimmed_mean(g[sel][:, cols], len(sel), 2 * f), want['out'], rtol=RTOL, atol=ATOL)


# ---- GPU ------------------------------------------------------------------------------------------------------------
def attacked_on_the_gpu(name, large, eng):
    """The matrix the reference's defences saw, made by OUR attack and nothing else (bit for bit the reference's vector)."""
    case, g = BY_NAME[name], seeded(name, large)
    if case.get('attack'):
        m = case['attack']
        drift, _, _ = eng.drift_attack(g[:m], case['z'])
        drift = np.asarray(drift)
        assert np.array_equal(drift[large[name]['drift_cols']], large[name]['drift'])
        g[:m] = drift
    return g


@pytest.mark.gpu
@pytest.mark.parametrize('name', ALL)
def test_gpu_krum_and_distances_are_the_references_at_large_sizes(eng, large, name, monkeypatch):
    torch = pytest.importorskip('torch')
    case, want = BY_NAME[name], large[name]
    n, f = case['n'], case['f']
    for key in ('BYZ_GRAM_MODE', 'BYZ_GRAM_PLANES', 'BYZ_DEDUP'):
        monkeypatch.delenv(key, raising=False)          # the production dispatch, whatever an earlier test left behind
    g = torch.from_numpy(attacked_on_the_gpu(name, large, eng)).cuda()
    dist = eng.pairwise_distances(g).numpy()
    rows = want['dist_rows']
    assert rows_close(dist[rows], want['dist'], rows), name
    assert np.all(np.isinf(np.diag(dist))) and np.array_equal(dist, dist.T)
    # defences.py:23-42 with return_index=True: the reference's index, through the engine's own distances
    assert eng.krum(g, n, f, return_index=True) == int(want['index'])
    assert eng.krum_select(dist, n, f) == int(want['index'])
    assert torch.equal(eng.krum(g, n, f), g[int(want['index'])])
    if case.get('attack'):
        # the attack's rows are one vector: their mutual distances are exact zeros and their rows of the matrix identical
        m = case['attack']
        assert np.all(dist[:m, :m][~np.eye(m, dtype=bool)] == 0.0)
        assert np.array_equal(dist[0, m:], dist[m - 1, m:])


@pytest.mark.gpu
@pytest.mark.parametrize('name', BULYAN)
def test_gpu_bulyan_is_the_references_at_large_sizes(eng, large, name, monkeypatch):
    torch = pytest.importorskip('torch')
    case, want = BY_NAME[name], large[name]
    n, f = case['n'], case['f']
    for key in ('BYZ_GRAM_MODE', 'BYZ_GRAM_PLANES', 'BYZ_DEDUP', 'BYZ_BULYAN_RESCORE'):
        monkeypatch.delenv(key, raising=False)
    host = attacked_on_the_gpu(name, large, eng)
    g = torch.from_numpy(host).cuda()
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    eng.check()
    sel = sel.cpu().tolist() if hasattr(sel, 'cpu') else list(sel)
    assert sel == want['selection'].tolist(), 'first difference at pick %d' % next(
        i for i, (a, b) in enumerate(zip(sel, want['selection'].tolist())) if a != b)
    assert np.allclose(out.cpu().numpy()[want['out_cols']], want['out'], rtol=RTOL, atol=ATOL)
    # the drop-in module, host numpy in (what server.py:87 passes)
    from attacking_federate_learning_amd import defences
    assert np.allclose(defences.bulyan(host, n, f)[want['out_cols']], want['out'], rtol=RTOL, atol=ATOL)
    # Krum and Bulyan on ONE distance matrix (configs[4]'s flow)
    handle = eng.pairwise_distances(g)
    assert eng.krum_select(handle, n, f) == int(want['index'])
    assert list(eng.bulyan_select(handle, n, f)) == want['selection'].tolist()
