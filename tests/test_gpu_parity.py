"""Parity of the HIP path against the CPU oracle and the reference's golden vectors (needs an MI355X).

Everything goes through the C ABI (libbyzagg.so) via the drop-in modules or the Engine.
Tolerances: indices bit-exact; aggregated fp32 vectors |d| <= 1e-5 + 1e-5*|ref| (BASELINE.json north_star).
"""
import warnings

import numpy as np
import pytest

from oracle import faithful, ideal

pytestmark = pytest.mark.gpu

RTOL = ATOL = 1e-5


@pytest.fixture(scope='module')
def defences(eng):
    from attacking_federate_learning_amd import defences
    return defences


def close(a, b, rtol=RTOL, atol=ATOL):
    return np.allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol, equal_nan=True)


def gaussian(seed, n, d):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


def scaled(seed, n, d):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    return g * (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]


# ---- primitives -------------------------------------------------------------------------------------
def test_lane_exchange_patterns(eng):
    got = eng.lane_exchange_selftest()
    masks = [1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63]
    assert got.shape == (len(masks), 64)
    for row, m in zip(got, masks):
        assert row.tolist() == [lane ^ m for lane in range(64)], 'lane_xor(%d)' % m


# ---- golden vectors minted from the reference -------------------------------------------------------
def test_golden_no_defense(defences, golden):
    c = golden['nodef_7x130']
    assert close(defences.no_defense(c['G'], 7, 1), c['out'])


@pytest.mark.parametrize('case', ['krum_iid_10x257', 'krum_scaled_33x1000', 'krum_attacked_12x300',
                                  'krum_allsame_6x64', 'krum_f0_5x40'])
def test_golden_krum(defences, eng, golden, case):
    c = golden[case]
    g, f, n = c['G'], int(c['f']), len(c['G'])
    # selection on the reference's own distance matrix: bit-exact scores -> identical index
    assert eng.krum_select(c['dist'], n, f) == int(c['index'])
    # end to end (Gram distances)
    assert defences.krum(g, n, f, return_index=True) == int(c['index'])
    assert np.array_equal(defences.krum(g, n, f), c['out'])
    dist = defences._krum_create_distances(g).numpy()
    off = ~np.eye(n, dtype=bool)
    assert np.all(np.isinf(np.diag(dist)))
    assert np.allclose(dist[off], c['dist'][off], rtol=2e-6, atol=1e-6)
    assert np.array_equal(dist, dist.T)


def test_golden_krum_all_nan(defences, golden):
    c = golden['krum_allnan_4x8']
    assert defences.krum(c['G'], 4, int(c['f']), return_index=True) == -1


@pytest.mark.parametrize('case', ['tm_odd_11x97', 'tm_even_10x97', 'tm_100x64', 'tm_attacked_20x50',
                                  'tm_c0_9x33', 'tm_edge_ties_c1', 'tm_edge_ties_c2', 'tm_edge_ties_c3',
                                  'tm_edge_ties_c4', 'tm_kzero_6x20', 'tm_kneg_6x20'])
def test_golden_trimmed_mean(defences, golden, case):
    c = golden[case]
    got = defences.trimmed_mean(c['G'], len(c['G']), int(c['c']))
    assert got.dtype == np.float32 and got.shape == c['out'].shape
    assert close(got, c['out']), (got, c['out'])


@pytest.mark.parametrize('case', ['bulyan_iid_11x200', 'bulyan_boundary_15x120', 'bulyan_scaled_40x500',
                                  'bulyan_attacked_23x150', 'bulyan_f0_6x30'])
def test_golden_bulyan(defences, eng, golden, case):
    c = golden[case]
    g, f, n = c['G'], int(c['f']), len(c['G'])
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    assert sel.tolist() == c['selection'].tolist()
    assert close(out, c['out'])
    assert close(defences.bulyan(g, n, f), c['out'])


@pytest.mark.parametrize('case', ['attack_5x300_z1.5', 'attack_24x100_z0.5', 'attack_3x64_z0'])
def test_golden_attack(eng, golden, case):
    from attacking_federate_learning_amd import malicious

    class User:
        def __init__(self, grads):
            self.grads, self.original_params, self.learning_rate = grads, None, None

    c = golden[case]
    users = [User(r.copy()) for r in c['G']]
    att = malicious.DriftAttack(float(c['z']))
    att.attack(users)
    # bit for bit since round 5 (the kernel is numpy's arithmetic, operation by operation)
    assert np.array_equal(att.grads_stdev, c['stored_stdev'])
    assert np.array_equal(att.grads_mean, c['stored_mean'])
    assert np.array_equal(users[0].grads, c['user0'])
    assert all(u.grads is users[0].grads for u in users) == bool(c['aliased'])


@pytest.mark.parametrize('name', ['NoDefense', 'Krum', 'TrimmedMean', 'Bulyan'])
def test_golden_dropin_rounds(defences, name):
    """Two rounds of `Server.defend` (server.py:86-90) on the matrices the UNMODIFIED reference main loop produced
    (recorded by tests/test_dropin_reference.py where /root/reference exists): the drop-in `defences.defend[...]` call
    of server.py:87 with its exact arguments, then the momentum step, must land on the reference's weights."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dropin_rounds.npz')
    z = np.load(path)
    mal_prop, momentum, learning_rate = float(z['mal_prop']), 0.9, 0.1
    weights = z['weights_start'].copy()
    velocity = np.zeros_like(weights)
    for e, key in enumerate(('round0_grads', '%s/round1_grads' % name)):
        users_grads = z[key]
        n = len(users_grads)
        before = users_grads.copy()
        current_grads = defences.defend[name](users_grads, n, int(n * mal_prop))        # server.py:87
        velocity = momentum * velocity - learning_rate * current_grads                      # server.py:89
        weights += velocity                                                                # server.py:90
        assert np.array_equal(users_grads, before)
        assert np.allclose(weights, z['%s/weights%d' % (name, e)], rtol=1e-5, atol=1e-6), (name, e)


def test_assertions_like_the_reference(defences):
    g = gaussian(1, 6, 10)
    with pytest.raises(AssertionError):
        defences.krum(g, 6, 3)
    with pytest.raises(AssertionError):
        defences.bulyan(g, 6, 3)
    assert isinstance(defences.krum(g, 6, 3, return_index=True), int)   # the assert is skipped (defences.py:24)
    assert set(defences.defend) == {'NoDefense', 'Krum', 'TrimmedMean', 'Bulyan'}


def test_krum_on_a_host_matrix_returns_a_view_like_the_reference(defences):
    """defences.py:42 returns `users_grads[minimal_error_index]`: a VIEW of the caller's matrix.  The host
    path does the same since round 6; with no score below 1e20 (all distances NaN) the index stays -1 = numpy's last row."""
    g = gaussian(2, 12, 300)
    row = defences.krum(g, 12, 2)
    idx = defences.krum(g, 12, 2, return_index=True)
    assert row.base is g and np.shares_memory(row, g) and np.array_equal(row, g[idx])
    assert idx == faithful.krum(g, 12, 2, return_index=True)
    g[idx, 0] = 123.0
    assert row[0] == 123.0                       # it IS the caller's row
    poisoned = g.copy()
    poisoned[:, 5] = np.nan
    last = defences.krum(poisoned, 12, 2)
    assert defences.krum(poisoned, 12, 2, return_index=True) == -1
    assert np.shares_memory(last, poisoned) and np.array_equal(last, poisoned[-1], equal_nan=True)


# ---- seeded random inputs against the oracle --------------------------------------------------------
@pytest.mark.parametrize('n,d', [(2, 1), (3, 7), (10, 79510), (100, 21840), (100, 79510), (129, 4097),
                                 (300, 20000), (1000, 3001)])
def test_distances_vs_fp64(eng, n, d):
    g = gaussian(1000 + n, n, d)
    got = eng.pairwise_distances(g).numpy()
    want = ideal.distance_matrix(g)
    off = ~np.eye(n, dtype=bool)
    rel = np.abs(got[off] - want[off]) / want[off]
    assert rel.max() < 1e-6, rel.max()          # the reference's own sdot noise is 5e-8 .. 4e-7
    assert np.array_equal(got, got.T) and np.all(np.isinf(np.diag(got)))


def test_distances_chunked_schedule_large_n(eng):
    """N > 2816 (>= 256 tiles) with a long K takes the chunked, ticketed Gram schedule and the bf16 x 3 arithmetic:
    fp64 reference on the GPU, identical rows must still give exact zeros and identical distance rows."""
    torch = pytest.importorskip('torch')
    n, d = 2900, 40000
    gen = torch.Generator(device='cuda').manual_seed(77)
    g = torch.randn((n, d), generator=gen, device='cuda', dtype=torch.float32)
    g[5] = g[2900 - 1]
    g[1700] = g[5]
    dist = torch.from_numpy(eng.pairwise_distances(g).numpy()).cuda()
    g64 = g.double()
    sq = (g64 * g64).sum(1)
    want = (sq[:, None] + sq[None, :] - 2.0 * (g64 @ g64.T)).clamp_min(0).sqrt()
    off = ~torch.eye(n, dtype=torch.bool, device='cuda')
    same = torch.zeros((n, n), dtype=torch.bool, device='cuda')
    for a in (5, 1700, n - 1):
        for b in (5, 1700, n - 1):
            same[a, b] = True
    mask = off & ~same
    rel = ((dist.double() - want).abs() / want)[mask]
    assert float(rel.max()) < 1e-6, float(rel.max())
    assert float(dist[5, 1700]) == 0.0 and float(dist[5, n - 1]) == 0.0 and float(dist[1700, n - 1]) == 0.0
    assert torch.equal(dist[5, mask[5]], dist[1700, mask[1700]])
    assert torch.equal(dist, dist.T)
    # a second call must reproduce the matrix bit for bit (fixed accumulation order through the tickets)
    again = torch.from_numpy(eng.pairwise_distances(g).numpy()).cuda()
    assert torch.equal(dist, again)


def test_identical_rows_are_found_before_the_gram(eng):
    """Under the attack a quarter of the rows are one vector (malicious.py:26-27).  From N = 512 the engine finds
    identical rows first (signature of sampled columns -> full bitwise verification) and runs the Gram over the
    unique rows only.  A row that differs from the group in ONE unsampled column has the group's signature and must be
    rejected by the verification; everything must equal the path without the shortcut."""
    import os
    n, d = 700, 40000
    rng = np.random.default_rng(123)
    g = scaled(124, n, d)
    group = np.sort(rng.choice(n, size=300, replace=False))
    g[group] = g[group[0]]
    near = int(np.setdiff1d(np.arange(n), group)[17])
    g[near] = g[group[0]]
    g[near, 3000] += 1.0                        # column 3000 lies in none of the eight sampled segments
    with_shortcut = eng.pairwise_distances(g).numpy()
    os.environ['BYZ_GRAM_DEDUP'] = '0'
    try:
        without = eng.pairwise_distances(g).numpy()
    finally:
        del os.environ['BYZ_GRAM_DEDUP']
    off = ~np.eye(n, dtype=bool)
    sub = with_shortcut[np.ix_(group, group)]
    assert np.all(sub[~np.eye(len(group), dtype=bool)] == 0.0)
    others = np.setdiff1d(np.arange(n), group)
    assert all(np.array_equal(with_shortcut[group[0], others], with_shortcut[i, others]) for i in group[1:])
    assert 0.9 < with_shortcut[near, group[0]] < 1.1 and with_shortcut[near, group[5]] == with_shortcut[near, group[0]]
    assert np.array_equal(with_shortcut, with_shortcut.T)
    # pairs of distinct rows meet the same operands in the same order either way
    rest = np.ix_(others, others)
    assert np.allclose(with_shortcut[rest][off[rest]], without[rest][off[rest]], rtol=1e-6, atol=0.0)
    assert np.allclose(with_shortcut[off], without[off], rtol=1e-5, atol=2e-2)   # near-duplicate: cancellation in d^2
    want = ideal.distance_matrix(g)
    far = want > 1.5
    assert np.allclose(with_shortcut[far], want[far], rtol=2e-6)
    # selection on top of it: the copies tie exactly and the reference's visit order decides, with or without
    f = 150
    assert eng.krum_select(with_shortcut, n, f) == eng.krum_select(without, n, f) == \
        faithful.krum_pick(with_shortcut, faithful.visit_order(n), n, f)
    assert np.array_equal(eng.bulyan_select(with_shortcut, n, f), eng.bulyan_select(without, n, f))


def test_gram_over_very_few_unique_rows(eng):
    """N >= 512 rows that are all one vector, or two vectors: the Gram runs over 1 or 2 rows and is expanded."""
    n, d = 600, 5000
    base = gaussian(31, 2, d)
    g = np.tile(base[0], (n, 1))
    dist = eng.pairwise_distances(g).numpy()
    assert np.all(dist[~np.eye(n, dtype=bool)] == 0.0)
    assert eng.krum_select(dist, n, 100) == 1                      # every score ties: visit order 1, 0, 2, ...
    g[::3] = base[1]
    dist = eng.pairwise_distances(g).numpy()
    want = np.float32(np.linalg.norm(base[0].astype(np.float64) - base[1].astype(np.float64)))
    same = (np.arange(n)[:, None] % 3 == 0) == (np.arange(n)[None, :] % 3 == 0)
    off = ~np.eye(n, dtype=bool)
    assert np.all(dist[same & off] == 0.0)
    assert np.allclose(dist[~same], want, rtol=1e-5) and len(np.unique(dist[~same])) == 1


def test_identical_rows_have_zero_distance_and_tie_exactly(eng):
    g = gaussian(7, 50, 33333)
    g[:12] = g[3]
    dist = eng.pairwise_distances(g).numpy()
    assert np.all(dist[:12, :12][~np.eye(12, dtype=bool)] == 0.0)
    assert all(np.array_equal(dist[0, 12:], dist[i, 12:]) for i in range(12))
    # every copy has the same score; the reference's visit order 1, 0, 2, ... decides
    idx = eng.krum_select(dist, 50, 12)
    want = faithful.krum_pick(dist, faithful.visit_order(50), 50, 12)
    assert idx == want


# ---- N <= 128 (csrc/krum_small.hip, the default there) next to the general path ----------------------------------
def _small_family(n, d, family, seed):
    rng = np.random.default_rng(seed)
    g = scaled(seed, n, d)
    if family == 'attack':          # the first quarter of the rows are one vector (malicious.py:26-27)
        m = max(2, n // 4)
        g[:m] = (g[:m].mean(axis=0) - 1.5 * g[:m].std(axis=0)).astype(np.float32)
    elif family == 'near':          # two rows that nearly coincide, one exact copy
        g[5] = g[3] + np.float32(1e-4) * rng.standard_normal(d).astype(np.float32)
        g[7] = g[2]
    elif family == 'tiny':
        g *= np.float32(1e-6)
    elif family == 'mixed':         # rows on very different scales
        g *= (10.0 ** rng.integers(-6, 6, size=n)).astype(np.float32)[:, None]
    return g


@pytest.mark.parametrize('n,d,f,family', [(10, 204, 2, 'scaled'), (33, 129, 8, 'scaled'), (64, 100, 10, 'scaled'),
                                          (5, 7, 1, 'scaled'), (100, 21840, 24, 'scaled'), (100, 79510, 24, 'scaled'),
                                          (128, 4099, 30, 'scaled'), (128, 128, 31, 'scaled'), (97, 8190, 24, 'attack'),
                                          (100, 79510, 24, 'attack'), (40, 5000, 9, 'near'), (100, 79510, 24, 'near'),
                                          (50, 3000, 12, 'tiny'), (50, 3000, 12, 'mixed'), (2, 300, 0, 'scaled'),
                                          (128, 40000, 31, 'attack'), (100, 255, 24, 'near'), (128, 127, 20, 'attack')])
def test_small_krum_path_next_to_the_general_path(eng, monkeypatch, n, d, f, family):
    """Both implementations of Krum for N <= 128 on the same rows: distances against the norm of the fp32 difference
    (defences.py:20) in fp64, the index against the reference's loop run on each path's OWN distance matrix, exact zeros
    and bitwise equal distance rows for identical rows, and the two paths against each other."""
    g = _small_family(n, d, family, 900 + n + d % 97)
    want = np.empty((n, n))
    for i in range(n):
        diff = (g[i][None, :] - g).astype(np.float32).astype(np.float64)
        want[i] = np.sqrt((diff * diff).sum(axis=1))
    off = ~np.eye(n, dtype=bool)
    got = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('BYZ_KRUM_SMALL', mode)
        dm = eng.pairwise_distances(g).numpy()
        idx = eng.krum(g, n, f, return_index=True)
        row = eng.krum(g, n, f)
        eng.check()
        assert np.all(np.isinf(np.diag(dm))) and np.array_equal(dm, dm.T)
        zero = want[off] == 0
        assert np.all(dm[off][zero] == 0.0)
        rel = np.abs(dm[off][~zero] - want[off][~zero]) / want[off][~zero]
        assert rel.max() < 1e-6, (mode, rel.max())
        assert idx == faithful.krum_pick(dm, faithful.visit_order(n), n, f), mode
        assert np.array_equal(row, g[idx])
        got[mode] = (dm, idx)
    assert got['0'][1] == got['1'][1]
    if family == 'attack':
        m = max(2, n // 4)
        dm = got['1'][0]
        assert all(np.array_equal(dm[0, m:], dm[i, m:]) for i in range(1, m))


def test_small_krum_path_covers_selection_and_bulyan(eng, monkeypatch, golden):
    """The entry points that share the N <= 128 kernels: krum_select on a given matrix (scores as the reference's sum()
    forms them), Bulyan end to end, and the golden Krum cases through the GENERAL path too (it stays reachable)."""
    rng = np.random.default_rng(77)
    for n, f in ((2, 0), (3, 1), (17, 4), (100, 24), (128, 31)):
        pts = rng.standard_normal((n, 9)).astype(np.float32)
        dist = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)).astype(np.float32)
        dist[np.arange(n), np.arange(n)] = np.inf
        dist[0, 1] = dist[1, 0]
        want = faithful.krum_pick(dist, faithful.visit_order(n), n, f)
        for mode in ('0', '1'):
            monkeypatch.setenv('BYZ_KRUM_SMALL', mode)
            assert eng.krum_select(dist, n, f) == want, (n, f, mode)
    g = scaled(4242, 43, 3000)
    monkeypatch.setenv('BYZ_KRUM_SMALL', '0')
    a, sel_a = eng.bulyan(g, 43, 10, return_selection=True)
    monkeypatch.setenv('BYZ_KRUM_SMALL', '1')
    b, sel_b = eng.bulyan(g, 43, 10, return_selection=True)
    assert np.array_equal(np.asarray(sel_a), np.asarray(sel_b)) and close(a, b)
    monkeypatch.setenv('BYZ_KRUM_SMALL', '0')
    for case in ('krum_iid_10x257', 'krum_scaled_33x1000', 'krum_attacked_12x300', 'krum_allsame_6x64', 'krum_f0_5x40'):
        c = golden[case]
        n, f = len(c['G']), int(c['f'])
        assert eng.krum_select(c['dist'], n, f) == int(c['index'])
        assert eng.krum(c['G'], n, f, return_index=True) == int(c['index'])


@pytest.mark.parametrize('n,d,f', [(40, 500, 9), (100, 3000, 24), (300, 2000, 70), (700, 1200, 168)])
def test_clients_with_infinite_gradients_are_never_chosen(eng, n, d, f):
    """Byzantine clients may send anything, +inf included.  In the reference every distance to such a client is +inf
    (np.linalg.norm of a difference with an infinite entry, defences.py:20), sorts last in every row and stays outside the
    summed prefix, and the client's own score is inf: Krum and Bulyan carry on with the finite clients.  Here the Gram identity
    turns those distances into NaN, which sort last just the same: same Krum index, same Bulyan selection and vector as the
    oracle on the same gradients (N <= 128: csrc/krum_small.hip; above: gram.hip + select.hip, whose running sums count the
    non-finite entries instead of adding them)."""
    g = scaled(5100 + n, n, d)
    for bad in (1, n // 3, n - 2):
        g[bad, 7] = np.inf
        g[bad, d - 1] = -np.inf
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        want_idx = faithful.krum(g, n, f, return_index=True)
    assert eng.krum(g, n, f, return_index=True) == want_idx
    # Bulyan on a given matrix: the reference's loop on the engine's own distances, with NaN read as the +inf the reference
    # has there (the Gram identity's last bits differ from np.linalg.norm's, so the comparison is on ONE matrix)
    dist = eng.pairwise_distances(g).numpy()
    assert not np.isfinite(dist[1, 0]) and not np.isfinite(dist[0, n - 2])
    want_sel = faithful.bulyan_selection(np.where(np.isnan(dist), np.float32(np.inf), dist), n, f)
    assert eng.bulyan_select(dist, n, f).tolist() == list(want_sel)
    assert not ({1, n // 3, n - 2} & set(want_sel))
    # ... and end to end: theta distinct finite clients, and the reference's trimmed mean over them
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    sel = sel.tolist()
    assert len(set(sel)) == n - 2 * f and not ({1, n // 3, n - 2} & set(sel))
    assert close(out, faithful.trimmed_mean(g[sel], len(sel), 2 * f))


@pytest.mark.parametrize('n,f', [(2, 0), (5, 1), (64, 15), (100, 24), (128, 31), (333, 80), (1000, 240)])
def test_krum_selection_is_bit_exact_given_distances(eng, n, f):
    rng = np.random.default_rng(2000 + n)
    pts = rng.standard_normal((n, 24)).astype(np.float32)
    dist = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)).astype(np.float32)
    np.fill_diagonal(dist, np.inf)
    want = faithful.krum_pick(dist, faithful.visit_order(n), n, f)
    assert eng.krum_select(dist, n, f) == want


@pytest.mark.parametrize('n,d,f', [(100, 79510, 24), (100, 21840, 24), (256, 10000, 60)])
def test_krum_end_to_end_margin_protocol(eng, n, d, f):
    g = scaled(3000 + n, n, d)
    idx = eng.krum(g, n, f, return_index=True)
    want, margin, scores = ideal.krum_index(ideal.distance_matrix(g), n, f, with_margin=True)
    tau = 16 * np.finfo(np.float32).eps
    if margin > tau:
        assert idx == want
    else:
        assert scores[idx] <= scores[want] * (1 + tau)
    assert np.array_equal(eng.krum(g, n, f), g[idx])


@pytest.mark.parametrize('n', [1, 2, 3, 10, 63, 64, 65, 100, 127, 128, 129, 500, 1000, 1024])
@pytest.mark.parametrize('d', [1, 33, 1000])
def test_trimmed_mean_register_kernel(eng, n, d):
    g = gaussian(4000 + n * 7 + d, n, d)
    c = n // 5
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        want = ideal.trimmed_mean(g, c)
    assert close(eng.trimmed_mean(g, n, c), want)


@pytest.mark.parametrize('n,c', [(7, 2), (8, 1), (40, 9), (100, 20), (257, 64)])
def test_trimmed_mean_heavy_ties_match_the_reference_rule(eng, n, c):
    # half-integer data: many exact +a / -a ties around the median, the stable row order decides
    g = (np.round(gaussian(5000 + n, n, 300) * 2) / 2).astype(np.float32)
    want = faithful.trimmed_mean(g, n, c)
    assert close(eng.trimmed_mean(g, n, c), want)


@pytest.mark.parametrize('n,d', [(1025, 64), (1500, 130), (2080, 257), (2561, 40), (4096, 36), (5200, 50), (5632, 17),
                                 (6000, 20), (8192, 9), (8193, 6), (10000, 7), (16384, 5)])
def test_trimmed_mean_general_kernel(eng, n, d):
    g = gaussian(6000 + n, n, d)
    c = n // 4
    assert close(eng.trimmed_mean(g, n, c), ideal.trimmed_mean(g, c))


@pytest.mark.parametrize('n,d', [(200, 50), (1000, 130), (2080, 48), (2561, 20), (5200, 33)])
def test_trimmed_mean_general_kernel_on_its_own(eng, monkeypatch, n, d):
    """BYZ_TM_RING=0: the general selection (median_window.hip, MODE 0) for every tile -- by default it only sees the few
    tiles the ring selection of window_lean.hip hands back, so its other instantiations get their own run here."""
    monkeypatch.setenv('BYZ_TM_RING', '0')
    g = scaled(6100 + n, n, d)
    for c in (n // 4, 1, n - 2):
        assert close(eng.trimmed_mean(g, n, c), ideal.trimmed_mean(g, c))
    assert eng.trimmed_mean_redone() == 0      # (no ring selection ran: nothing was handed back)


@pytest.mark.parametrize('n,d', [(100, 4096), (300, 2048), (640, 1024)])
def test_gram_register_staging_equals_lds_dma(eng, monkeypatch, n, d):
    """BYZ_GRAM_NO_DMA=1 stages the Gram's operand tiles through registers (the path rows that are not 16-byte aligned
    take anyway) instead of LDS-DMA: same arithmetic, same k order -- the distances must be bit for bit the same.  (The
    fp32-input MFMA on both sides: the bf16 x 3 split exists only on the LDS-DMA path.)"""
    monkeypatch.setenv('BYZ_KRUM_SMALL', '0')
    monkeypatch.setenv('BYZ_GRAM_MODE', 'exact')
    g = scaled(6200 + n, n, d)
    monkeypatch.delenv('BYZ_GRAM_NO_DMA', raising=False)
    want = eng.pairwise_distances(g).numpy()
    monkeypatch.setenv('BYZ_GRAM_NO_DMA', '1')
    got = eng.pairwise_distances(g).numpy()
    assert np.array_equal(got, want)


@pytest.mark.parametrize('n,m', [(300, 72), (1000, 240), (130, 129)])
def test_trimmed_mean_with_many_identical_rows(eng, n, m):
    """More than 64 clients submit the same vector (the attack's normal case): a bucket then holds more equal values
    than the 64-lane sort takes and the kernel must fall back to the probing search."""
    g = scaled(5100 + n, n, 257)
    g[:m] = faithful.drift_vector(g[:m].copy(), 1.5)
    c = n // 5
    assert close(eng.trimmed_mean(g, n, c), ideal.trimmed_mean(g, c))
    assert close(eng.trimmed_mean(g[:, :64], n, c), faithful.trimmed_mean(g[:, :64], n, c))


def test_trimmed_mean_with_outliers(eng):
    """+-1e30 outliers squeeze every other value into one bucket of the first histogram."""
    g = gaussian(5200, 1000, 300)
    g[3, :] = 1e30
    g[700, ::2] = -1e30
    g[11, 5] = np.inf
    want = ideal.trimmed_mean(g, 200)
    assert close(eng.trimmed_mean(g, 1000, 200), want)


def test_trimmed_mean_row_index_orders_the_rows(eng):
    g = (np.round(gaussian(61, 60, 200)) ).astype(np.float32)   # integers: ties everywhere
    order = np.random.default_rng(3).permutation(60)[:41].astype(np.int32)
    want = faithful.trimmed_mean(g[order], 41, 10)
    gd = eng.to_device(g)
    got = eng.trimmed_mean(gd, 60, 10, row_index=order).numpy()
    assert close(got, want)


@pytest.mark.parametrize('n,d,f', [(11, 50, 2), (40, 3000, 9), (100, 20000, 24), (301, 5000, 74),
                                   (1000, 2000, 240)])
def test_bulyan_vs_fp64_oracle(eng, n, d, f):
    g = scaled(7000 + n, n, d)
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    dist = ideal.distance_matrix(g)
    want_sel, margins = ideal.bulyan_selection(dist, n, f, with_margins=True)
    tau = 16 * np.finfo(np.float32).eps
    sel = sel.tolist()
    assert len(sel) == n - 2 * f and len(set(sel)) == len(sel)
    first_noisy = int(np.argmax(margins <= tau)) if np.any(margins <= tau) else len(sel)
    assert sel[:first_noisy] == want_sel[:first_noisy]
    if first_noisy == len(sel):
        assert close(out, ideal.trimmed_mean(g[want_sel], 2 * f))
    # the protocol's second clause (tests/test_gpu_scale.py::margin_protocol): the engine's selection replayed
    # in fp64 IN ITS OWN STATE -- every pick, also those after the first contested one, within tau of the optimum; a pick that
    # is not the fp64 argmin is a contested one; the count of contested picks against a ceiling (measured: 0 .. 4 at these sizes)
    from oracle import scale
    excess, margin, argmin = scale.replay_selection(np.ascontiguousarray(dist, dtype=np.float32), n, f, sel, mode='ideal')
    assert float(excess.max()) <= tau, (float(excess.max()), tau)
    assert np.all(margin[argmin != np.asarray(sel, dtype=np.int32)] <= tau)
    assert int((margin <= tau).sum()) <= 12, int((margin <= tau).sum())
    # whatever was picked, the second stage must be the reference's trimmed mean of exactly those rows
    assert close(out, ideal.trimmed_mean(g[sel], 2 * f))


def test_bulyan_selection_matches_fp64_given_the_same_distances(eng):
    rng = np.random.default_rng(77)
    pts = rng.standard_normal((200, 16)).astype(np.float32)
    dist = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)).astype(np.float32)
    np.fill_diagonal(dist, np.inf)
    for f in (0, 1, 20, 49):
        assert eng.bulyan_select(dist, 200, f).tolist() == ideal.bulyan_selection(dist, 200, f)


def test_bulyan_under_the_drift_attack_keeps_exact_ties(eng):
    n, d, f = 43, 4000, 10
    g = scaled(81, n, d)
    g[:f] = faithful.drift_vector(g[:f].copy(), 1.5)
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    want_out, want_sel = faithful.bulyan(g, n, f, return_selection=True)
    assert sel.tolist() == want_sel
    assert close(out, want_out)


@pytest.mark.parametrize('m,d,z', [(1, 10, 1.5), (24, 79510, 1.5), (240, 5000, 0.7), (5, 1 << 20, 1.5)])
def test_drift_attack_statistics(eng, m, d, z):
    g = gaussian(8000 + m, m, d) * 2 + 0.5
    drift, mean, std = eng.drift_attack(g, z)
    want_mean, want_std = faithful.attack_statistics(g)
    assert np.array_equal(mean, want_mean) and np.array_equal(std, want_std)
    assert np.array_equal(drift, faithful.drift_vector(g, z))


def test_no_defense_sizes(eng):
    for n, d in [(1, 1), (10, 79510), (100, 21840), (1000, 4096), (7, 1 << 20)]:
        g = gaussian(9000 + n, n, d)
        assert close(eng.no_defense(g), np.mean(g, axis=0))


def test_server_update_is_bit_exact(eng):
    rng = np.random.default_rng(5)
    w, v, a = (rng.standard_normal(100003).astype(np.float32) for _ in range(3))
    wd, vd, ad = eng.to_device(w), eng.to_device(v), eng.to_device(a)
    eng.server_update(wd, vd, ad, 0.9, 0.1)
    eng.synchronize()
    v2 = np.float32(0.9) * v - np.float32(0.1) * a      # server.py:89
    assert np.array_equal(vd.numpy(), v2) and np.array_equal(wd.numpy(), w + v2)


# ---- the steps either side of the path ---------------------------------------------
@pytest.mark.parametrize('case', ['backdoor_300_z1.5', 'backdoor_1000_z0.5_faded_lr', 'backdoor_64_z0'])
def test_golden_backdoor_hook_is_bit_exact(eng, golden, case):
    c = golden[case]
    lr, z = float(c['lr']), float(c['z'])
    assert np.array_equal(eng.backdoor_initial_params(c['params'], c['mean'], lr), c['start'])
    assert np.array_equal(eng.backdoor_clip(c['mean'], c['stdev'], c['params'], c['mal'], lr, z), c['out'])


@pytest.mark.parametrize('d', [1, 3, 4, 1021, 79510, (1 << 20) + 5])
def test_backdoor_hook_sizes_and_nan(eng, d):
    rng = np.random.default_rng(4400 + d % 1000)
    mean, params = (rng.standard_normal(d).astype(np.float32) for _ in range(2))
    stdev = np.abs(rng.standard_normal(d)).astype(np.float32)
    mal = (params + 0.5 * rng.standard_normal(d)).astype(np.float32)
    if d > 4:   # np.clip propagates NaN from the value and from either bound
        mal[1], stdev[2], mean[3] = np.nan, np.nan, np.inf
    lr, z = 0.1 * 10 / 17, 1.5
    with np.errstate(invalid='ignore'):
        want = faithful.backdoor_attack_grads(mean, stdev, params, lr, z, mal)
    got = eng.backdoor_clip(mean, stdev, params, mal, lr, z)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(eng.backdoor_initial_params(params, mean, lr), faithful.backdoor_initial_params(params, lr, mean),
                          equal_nan=True)


def test_backdoor_attack_class_drives_the_hook(eng):
    """malicious.Attack.attack -> BackdoorAttack._attack_grads, as main.py:54-68 would call it."""
    from attacking_federate_learning_amd.backdoor import BackdoorAttack

    class User:
        def __init__(self, grads):
            self.grads, self.original_params, self.learning_rate = grads, None, None

    rng = np.random.default_rng(77)
    g = rng.standard_normal((6, 500)).astype(np.float32)
    users = [User(r.copy()) for r in g]
    users[0].original_params = rng.standard_normal(500).astype(np.float32)
    users[0].learning_rate = 0.1
    trained = (users[0].original_params + rng.standard_normal(500)).astype(np.float32)
    starts = []
    att = BackdoorAttack(1.5, train_malicious_network=lambda start: (starts.append(np.asarray(start).copy()), trained)[1])
    att.attack(users)
    mean, stdev = faithful.attack_statistics(g)
    assert close(att.grads_mean, mean) and close(att.grads_stdev, stdev, atol=1e-6)
    # the hook's arithmetic is exact given the statistics the engine produced
    want = faithful.backdoor_attack_grads(att.grads_mean, att.grads_stdev, users[0].original_params, 0.1, 1.5, trained)
    assert np.array_equal(users[0].grads, want) and all(u.grads is users[0].grads for u in users)
    assert np.array_equal(starts[0], faithful.backdoor_initial_params(users[0].original_params, 0.1, att.grads_mean))


def test_golden_gradient_assembly(eng, golden):
    from attacking_federate_learning_amd.assembly import GradientMatrix
    c = golden['assemble_4x204']
    # host vectors (what usr.grads is in the reference) ...
    gm = GradientMatrix(4, 204, engine=eng)

    class User:
        pass
    users = []
    for u in range(4):
        usr = User()
        usr.grads = np.concatenate([c['u%d_t%d' % (u, t)].flatten() for t in range(5)])
        users.append(usr)
    gm.collect_gradients(users)
    assert np.array_equal(gm.numpy(), c['G'])
    # ... and per-parameter device tensors, concatenated on the GPU
    gm2 = GradientMatrix(4, 204, engine=eng)
    for u in range(4):
        gm2.set_row(u, [eng.to_device(c['u%d_t%d' % (u, t)]) for t in range(5)])
    eng.synchronize()
    assert np.array_equal(gm2.numpy(), c['G'])


def test_gradient_assembly_from_torch_parameters(eng):
    """Many tensors of awkward sizes and alignments (more than one launch's table), straight from torch."""
    torch = pytest.importorskip('torch')
    from attacking_federate_learning_amd.assembly import GradientMatrix
    rng = np.random.default_rng(12)
    sizes = [1, 3, 784 * 100, 100, 7, 100 * 10, 10, 5, 4096, 33] * 4          # 40 tensors > 32 per launch
    d = sum(sizes)
    gm = GradientMatrix(3, d, engine=eng, torch_device='cuda')
    want = np.empty((3, d), dtype=np.float32)
    for u in range(3):
        pool = torch.from_numpy(rng.standard_normal(d + 64).astype(np.float32)).cuda()
        tensors, off = [], u      # misaligned views of one pool: every alignment case of the copy kernel
        for n in sizes:
            tensors.append(pool[off:off + n])
            off += n
        gm.set_row(u, tensors)
        faithful.assemble_row(want, u, [t.cpu().numpy() for t in tensors])
    torch.cuda.synchronize()
    assert np.array_equal(gm.numpy(), want)
    with pytest.raises(ValueError):
        gm.set_row(0, [tensors[0]])                       # too few values for the row
    # every client at once, from batched per-parameter gradients (n_clients, *shape)
    shapes = [(100, 784), (100,), (10, 100), (10,), (3,), (1,)]
    batched = [torch.from_numpy(rng.standard_normal((5,) + sh).astype(np.float32)).cuda() for sh in shapes]
    gm5 = GradientMatrix(5, sum(int(np.prod(sh)) for sh in shapes), engine=eng, torch_device='cuda')
    gm5.set_all(batched)
    want5 = np.empty(gm5.shape, dtype=np.float32)
    for u in range(5):
        faithful.assemble_row(want5, u, [b[u].cpu().numpy() for b in batched])
    assert np.array_equal(gm5.numpy(), want5)
    # the assembled matrix feeds the defences directly
    assert close(eng.no_defense(gm.data).cpu().numpy(), np.mean(want, axis=0))


def test_batched_client_step_fills_the_device_matrix(eng, golden):
    """all clients' gradients in one batched pass on the GPU, written straight into the
    device-resident matrix, which the defences then consume -- against the golden vectors from the reference's
    own User.step and, at the reference's batch size, against the per-client loop of the oracle."""
    torch = pytest.importorskip('torch')
    from oracle import clients as oracle_clients
    from attacking_federate_learning_amd.assembly import GradientMatrix
    from attacking_federate_learning_amd.clients import collect_batched
    c = golden['clients_mnist_3x5']
    net = oracle_clients.MnistNet().cuda()
    gm = GradientMatrix(3, 79510, engine=eng, torch_device='cuda')
    collect_batched(gm, net, c['weights'], torch.from_numpy(c['data']).view(3, 5, 784).cuda(),
                    torch.from_numpy(c['target']).cuda())
    rows = gm.numpy()
    w1 = rows[:, :78400].reshape(3, 100, 784)
    assert np.allclose(w1[:, (0, 57), :], c['fc1_weight_rows'], rtol=1e-5, atol=1e-6)
    assert np.allclose(rows[:, 78400:], c['tail'], rtol=1e-5, atol=1e-6)
    assert np.allclose(np.sqrt((rows.astype(np.float64) ** 2).sum(axis=1)), c['row_norms'], rtol=1e-5)
    # N = 20 clients, batch 83 (main.py's default): the loop of the oracle on the host vs one batched pass
    rng = np.random.default_rng(31)
    n, batch = 20, 83
    weights = (0.05 * rng.standard_normal(79510)).astype(np.float32)
    data = rng.standard_normal((n, batch, 784)).astype(np.float32)
    target = rng.integers(0, 10, size=(n, batch))
    want = oracle_clients.all_client_gradients(oracle_clients.MnistNet(), weights, torch.from_numpy(data),
                                               torch.from_numpy(target), flatten_input=False)
    gm = GradientMatrix(n, 79510, engine=eng, torch_device='cuda')
    collect_batched(gm, net, torch.from_numpy(weights).cuda(), torch.from_numpy(data).cuda(),
                    torch.from_numpy(target).cuda())
    assert np.allclose(gm.numpy(), want, rtol=1e-4, atol=1e-6)
    # and the round goes on without leaving the GPU: Krum on the assembled matrix picks the oracle's client
    f = 4
    assert eng.krum(gm.data, n, f, return_index=True) == ideal.krum_index(ideal.distance_matrix(want), n, f)


@pytest.mark.parametrize('defence', ['Krum', 'TrimmedMean', 'Bulyan', 'NoDefense'])
def test_device_server_rounds_follow_the_host_loop(eng, defence):
    """Three whole rounds on the GPU (batched client step -> drift attack -> defence -> momentum step, server.py:54-56,
    81-90 and main.py:66-71) against the same rounds done on the host by the oracle."""
    torch = pytest.importorskip('torch')
    from oracle import clients as oracle_clients
    from attacking_federate_learning_amd.server import DeviceServer
    rng = np.random.default_rng(55)
    n, batch, mal_prop, lr, momentum, z = 15, 16, 0.2, 0.1, 0.9, 1.5
    f = int(n * mal_prop)
    weights0 = (0.05 * rng.standard_normal(79510)).astype(np.float32)
    data = rng.standard_normal((3, n, batch, 784)).astype(np.float32)
    target = rng.integers(0, 10, size=(3, n, batch))
    net = oracle_clients.MnistNet().cuda()
    srv = DeviceServer(n, weights0, mal_prop, lr, momentum, engine=eng)
    w, v = weights0.copy(), np.zeros_like(weights0)
    host_net = oracle_clients.MnistNet()
    for r in range(3):
        srv.collect_batched(net, torch.from_numpy(data[r]).cuda(), torch.from_numpy(target[r]).cuda())
        srv.attack(f, z)
        srv.defend(defence, r)
        # the same round on the host
        g = oracle_clients.all_client_gradients(host_net, w, torch.from_numpy(data[r]), torch.from_numpy(target[r]),
                                                flatten_input=False)
        g[:f] = faithful.drift_vector(g[:f].copy(), z)
        agg = {'Krum': faithful.krum, 'TrimmedMean': faithful.trimmed_mean, 'Bulyan': faithful.bulyan,
               'NoDefense': faithful.no_defense}[defence](g, n, f)
        v = np.float32(momentum) * v - np.float32(lr) * agg        # server.py:89
        w = w + v                                                  # server.py:90
        got = srv.current_weights.cpu().numpy()
        assert np.allclose(got, w, rtol=1e-4, atol=1e-6), 'round %d' % r
    assert np.allclose(srv.velocity.cpu().numpy(), v, rtol=1e-4, atol=1e-6)


# ---- device-resident (torch) inputs: zero-copy path -------------------------------------------------
def test_torch_device_tensors(eng):
    torch = pytest.importorskip('torch')
    g = scaled(99, 64, 5000)
    gt = torch.from_numpy(g).cuda()
    assert close(eng.no_defense(gt).cpu().numpy(), np.mean(g, axis=0))
    assert close(eng.trimmed_mean(gt, 64, 12).cpu().numpy(), ideal.trimmed_mean(g, 12))
    idx = eng.krum(gt, 64, 15, return_index=True)
    assert idx == ideal.krum_index(ideal.distance_matrix(g), 64, 15)
    assert torch.equal(eng.krum(gt, 64, 15), gt[idx])
    out, sel = eng.bulyan(gt, 64, 15, return_selection=True)
    assert sel.cpu().tolist() == ideal.bulyan_selection(ideal.distance_matrix(g), 64, 15)
    assert close(out.cpu().numpy(), ideal.trimmed_mean(g[sel.cpu().numpy()], 30))
    # a strided view: leading dimension != width
    wide = torch.from_numpy(scaled(98, 32, 600)).cuda()
    view = wide[:, 100:357]
    assert close(eng.trimmed_mean(view, 32, 6).cpu().numpy(), ideal.trimmed_mean(view.cpu().numpy(), 6))
    drift, mean, std = eng.drift_attack(gt[:15].clone(), 1.5, write_back=False)
    assert close(drift.cpu().numpy(), faithful.drift_vector(g[:15].copy(), 1.5))


# ---- full BASELINE sizes: size-independent properties -----------------------------------------------
def test_config3_trimmed_mean_properties(eng):
    """N=1000, D=1e6, trim 20%: spot-check columns against the oracle, plus invariances of the rule."""
    torch = pytest.importorskip('torch')
    n, d, c = 1000, 1_000_000, 200
    gen = torch.Generator(device='cuda').manual_seed(1236)
    g = torch.randn((n, d), generator=gen, device='cuda', dtype=torch.float32)
    out = eng.trimmed_mean(g, n, c)
    cols = np.random.default_rng(0).choice(d, 256, replace=False)
    sub = g[:, torch.from_numpy(cols).cuda()].cpu().numpy()
    assert close(out.cpu().numpy()[cols], ideal.trimmed_mean(sub, c))
    # permuting the clients cannot change a continuous-data result beyond summation order
    perm = torch.randperm(n, device='cuda', generator=gen)
    gp = g[perm].contiguous()
    out_p = eng.trimmed_mean(gp, n, c)
    differs = torch.nonzero(~torch.isclose(out, out_p, rtol=1e-5, atol=1e-5)).flatten()
    # ... except where |x - med| ties exactly across the median at the window edge: there the reference's
    # stable sort makes the row order matter (about 2^-23 per column for continuous data).  Every such
    # column must agree with the oracle in BOTH row orders.
    assert differs.numel() <= 8, differs.numel()
    if differs.numel():
        a, b = g[:, differs].cpu().numpy(), gp[:, differs].cpu().numpy()
        assert close(out[differs].cpu().numpy(), faithful.trimmed_mean(a, n, c))
        assert close(out_p[differs].cpu().numpy(), faithful.trimmed_mean(b, n, c))
    # the result lies between the kept window's extremes, hence inside the column's range
    assert torch.all(out <= g.max(dim=0).values) and torch.all(out >= g.min(dim=0).values)
    # translation equivariance: g + 4 rounds every value to a multiple of 2^-21, which may flip a near-tie
    # at a window edge (a 2a/k jump); such columns are rare and each must still agree with the oracle
    gs = g + 4.0
    out_s = eng.trimmed_mean(gs, n, c)
    moved = torch.nonzero(~torch.isclose(out_s, out + 4.0, rtol=1e-5, atol=2e-5)).flatten()
    assert moved.numel() <= d // 1000, moved.numel()
    if moved.numel():
        pick = moved[:64]
        assert close(out_s[pick].cpu().numpy(), ideal.trimmed_mean(gs[:, pick].cpu().numpy(), c))


def test_config2_krum_full_size_linearity(eng):
    torch = pytest.importorskip('torch')
    n, d, f = 100, 79510, 24
    g = torch.from_numpy(scaled(1235, n, d)).cuda()
    idx = eng.krum(g, n, f, return_index=True)
    # distances are homogeneous of degree 1: scaling all gradients by 2 cannot change the winner
    assert eng.krum(g * 2.0, n, f, return_index=True) == idx
    # and a common offset cancels in every difference
    assert eng.krum(g + 0.25, n, f, return_index=True) == idx
