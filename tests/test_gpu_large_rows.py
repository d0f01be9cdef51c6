"""More rows than the LDS-resident kernels hold: csrc/large_rows.hip (needs an MI355X).

The reference has no row limit (defences.py:23-70 are loops over Python lists); the kernels of select.hip / trimmed_mean.hip stop at
16,384 rows.  Beyond that the rows (columns) are sorted in global memory and the Bulyan loop decides batches of picks on exact fp64
scores, with every contender inside the rigorous rounding band scored again the reference's way and a batch cut where a guess fails.  Two halves:

  * the large path FORCED at sizes the C oracle recomputes completely (BYZ_SELECT_LARGE=1 / BYZ_TM_LARGE=1): the selection pick for pick
    against oracle/scale.py (defences.py:26-37, :59-68 restated in C) and against the production kernels, on contested data, twins, exact
    ties, other prefix lengths, non-finite and negative entries, the reference's KeyError; the trimmed mean against oracle.faithful;
  * the path at its own sizes (16,385 .. 20,001 rows): Krum's index against the oracle, Bulyan's selection through sampled picks
    (each one the reference's full scoring pass in the state of that pick), the trimmed mean against oracle.faithful.
"""
import ctypes

import numpy as np
import pytest

from oracle import faithful, scale

pytestmark = pytest.mark.gpu


def close(a, b, rtol=1e-5, atol=1e-5):
    return np.allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol, equal_nan=True)


def point_distances(seed, n, dim, identical=0, quantum=None):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    pts *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if identical:
        pts[:identical] = pts[:identical].mean(axis=0)
    sq = (pts.astype(np.float64) ** 2).sum(1)
    d = sq[:, None] + sq[None, :]
    d -= 2.0 * (pts.astype(np.float64) @ pts.astype(np.float64).T)
    d = np.sqrt(np.maximum(d, 0.0)).astype(np.float32)
    d = np.minimum(d, d.T)
    if quantum:
        d = (np.round(d / quantum) * quantum).astype(np.float32)
    if identical:
        d[:identical, :identical] = 0.0
        d[:identical, :] = d[0, :]
        d[:, :identical] = d[:, [0]]
    np.fill_diagonal(d, np.inf)
    return d


def selection_or_error(fn, *args):
    try:
        return np.asarray(fn(*args)).tolist()
    except KeyError:
        return KeyError


@pytest.fixture
def large(monkeypatch):
    monkeypatch.setenv('BYZ_SELECT_LARGE', '1')
    monkeypatch.setenv('BYZ_TM_LARGE', '1')
    return monkeypatch


# ---- the large path forced at sizes the oracle recomputes completely ---------------------------------------------------------------
@pytest.mark.parametrize('n,dim,identical,quantum', [
    (300, 4, 0, None),          # few dimensions: nearly every pick contested
    (777, 16, 0, None),
    (1500, 2000, 0, None),      # scores a rounding error apart
    (900, 16, 216, None),       # the attack's twins (malicious.py:26-27)
    (520, 8, 0, 0.25),          # exact ties between rows that are not twins: the visit order 1, 0, 2, ... decides
    (2049, 16, 0, None),        # one key more than a power of two
    (4100, 64, 984, None),      # more than one LDS chunk per row: the sort's global strides
])
def test_forced_large_selection_is_the_reference_selection(eng, large, n, dim, identical, quantum):
    dist = point_distances(5200 + n, n, dim, identical, quantum)
    f = int(0.24 * n)
    want = selection_or_error(scale.bulyan_selection, dist, n, f)
    got = selection_or_error(eng.bulyan_select, dist, n, f)
    assert got == want, next(i for i, (x, y) in enumerate(zip(got, want)) if x != y)
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)
    idx, picks = eng.krum_bulyan_select(dist, n, f)          # one sort for both (configs[4]'s round)
    assert np.asarray(picks).tolist() == want and idx == scale.krum_pick(dist, n, f)
    large.delenv('BYZ_SELECT_LARGE')
    assert np.asarray(eng.bulyan_select(dist, n, f)).tolist() == want      # ... and the production kernels agree
    eng.check()


def test_forced_large_row_sort_in_several_batches(eng, large):
    """The rows are sorted a batch at a time (the key scratch is bounded): three batches, the last one ragged."""
    n, f = 700, 160
    dist = point_distances(31, n, 5)
    large.setenv('BYZ_LARGE_SCRATCH_MB', '2')        # 1024 keys x 8 B per row: 256 rows per batch
    assert np.asarray(eng.bulyan_select(dist, n, f)).tolist() == scale.bulyan_selection(dist, n, f)
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)


def test_forced_large_on_a_matrix_that_contests_everything(eng, large):
    n, f = 600, 100
    dist = np.full((n, n), 3.0, dtype=np.float32)
    np.fill_diagonal(dist, np.inf)
    want = scale.bulyan_selection(dist, n, f)
    assert want[:4] == [1, 0, 2, 3]
    assert np.asarray(eng.bulyan_select(dist, n, f)).tolist() == want
    assert eng.bulyan_rescored() >= sum(n - t for t in range(n - 2 * f))      # every live row, at every pick


@pytest.mark.parametrize('users_delta,corrupted', [(0, 60), (0, 1), (-40, 50), (25, 30)])
def test_forced_large_with_other_prefix_lengths(eng, large, users_delta, corrupted):
    n = 420
    dist = point_distances(77, n, 6)
    users = n + users_delta
    assert selection_or_error(eng.bulyan_select, dist, users, corrupted) == selection_or_error(scale.bulyan_selection, dist, users, corrupted)
    assert eng.krum_select(dist, users, corrupted) == scale.krum_pick(dist, users, corrupted)


def test_forced_large_with_non_finite_and_negative_entries(eng, large):
    n, f = 512, 100
    dist = point_distances(91, n, 5)
    rng = np.random.default_rng(5)
    for _ in range(40):
        i, j = rng.integers(0, n, 2)
        if i != j:
            dist[i, j] = dist[j, i] = np.inf
    for _ in range(300):
        i, j = rng.integers(0, n, 2)
        if i != j:
            dist[i, j] = dist[j, i] = -abs(dist[i, j]) * 1e-3 if np.isfinite(dist[i, j]) else dist[i, j]
    assert selection_or_error(eng.bulyan_select, dist, n, f) == selection_or_error(scale.bulyan_selection, dist, n, f)
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)


def test_forced_large_when_the_reference_gives_up(eng, large):
    """No score below 1e20: the reference pops key -1 (KeyError, defences.py:65): at the first pick, and in the middle of the loop."""
    n, f = 400, 40
    huge = np.full((n, n), 1e19, dtype=np.float32)
    np.fill_diagonal(huge, np.inf)
    assert selection_or_error(scale.bulyan_selection, huge, n, f) is KeyError
    assert selection_or_error(eng.bulyan_select, huge, n, f) is KeyError
    mid = point_distances(15, n, 3)
    mid[50:, 50:] = np.inf
    assert selection_or_error(scale.bulyan_selection, mid, n, f) is KeyError
    assert selection_or_error(eng.bulyan_select, mid, n, f) is KeyError
    users, corrupted = 130, 40
    want = scale.bulyan_selection(mid, users, corrupted)
    assert sorted(want) == list(range(50))
    assert np.asarray(eng.bulyan_select(mid, users, corrupted)).tolist() == want
    assert eng.krum_select(huge, n, f) == -1 == scale.krum_pick(huge, n, f)


@pytest.mark.parametrize('batch', ['1', '3', '16', '32', '16 on one workgroup'])
def test_forced_large_batches_select_the_same(eng, large, batch):
    """BYZ_LARGE_BATCH: picks decided on the exact scores before their contenders are scored together, the batch cut where the
    reference's winner is not the guess.  Whatever the batch length, and whether the deciding kernel runs on the whole chip (a
    cooperative launch, two grid barriers per pick) or on one workgroup (BYZ_LARGE_COOP=0): the reference's selection -- on data that contests most picks
    (guesses fail), on twins and exact ties (they must not), with rows that always contend, and where the reference gives up in the
    middle of a batch."""
    if batch.endswith('on one workgroup'):
        large.setenv('BYZ_LARGE_COOP', '0')       # the deciding kernel without the cooperative launch (one workgroup walks all rows)
    large.setenv('BYZ_LARGE_BATCH', batch.split()[0])
    for seed, n, dim, identical, quantum in [(1, 300, 2, 0, None), (2, 900, 16, 216, None), (3, 520, 8, 0, 0.25), (4, 1500, 2000, 0, None)]:
        dist = point_distances(8000 + seed, n, dim, identical, quantum)
        f = int(0.24 * n)
        assert np.asarray(eng.bulyan_select(dist, n, f)).tolist() == scale.bulyan_selection(dist, n, f), (seed, batch)
    n, f = 400, 40
    mid = point_distances(15, n, 3)
    mid[50:, 50:] = np.inf
    assert selection_or_error(eng.bulyan_select, mid, n, f) is KeyError
    assert np.asarray(eng.bulyan_select(mid, 130, 40)).tolist() == scale.bulyan_selection(mid, 130, 40)
    odd = point_distances(91, 512, 5)
    rng = np.random.default_rng(5)
    for _ in range(200):
        i, j = rng.integers(0, 512, 2)
        if i != j:
            odd[i, j] = odd[j, i] = np.inf if _ % 5 == 0 else -abs(odd[i, j]) * 1e-3
    assert selection_or_error(eng.bulyan_select, odd, 512, 100) == selection_or_error(scale.bulyan_selection, odd, 512, 100)


@pytest.mark.parametrize('n,cols,c', [(50, 37, 10), (129, 8, 40), (1000, 23, 480), (2080, 9, 1920), (5000, 6, 2400)])
def test_forced_large_trimmed_mean(eng, large, n, cols, c):
    rng = np.random.default_rng(6100 + n)
    # quarter-integer data: exact +t / -t ties at the window edge, resolved by row order (defences.py:50)
    g = (np.round(rng.standard_normal((n, cols)) * 64) / 64).astype(np.float32)
    assert close(eng.trimmed_mean(eng.to_device(g), n, c).numpy(), faithful.trimmed_mean(g, n, c))
    g2 = rng.standard_normal((n, cols)).astype(np.float32)
    order = rng.permutation(n)[:n - c // 2].astype(np.int32)
    got = eng.trimmed_mean(eng.to_device(g2), n, c // 2, row_index=order).numpy()
    assert close(got, faithful.trimmed_mean(g2[order], len(order), c // 2))
    g2[3, 1] = np.nan          # np.median of a column with a NaN is NaN, and so is everything after it
    g2[n - 1, 2] = -np.nan
    got = eng.trimmed_mean(eng.to_device(g2), n, c).numpy()
    want = faithful.trimmed_mean(g2, n, c)
    assert np.isnan(got[1]) and np.isnan(got[2]) and close(np.delete(got, [1, 2]), np.delete(want, [1, 2]))
    large.setenv('BYZ_LARGE_SCRATCH_MB', '1')       # several batches of columns
    assert close(eng.trimmed_mean(eng.to_device(g), n, c).numpy(), faithful.trimmed_mean(g, n, c))


# ---- beyond 16,384 rows ------------------------------------------------------------------------------------------------------------
def test_limits_say_so(eng):
    from attacking_federate_learning_amd import _native
    lib = _native.load()
    a, b = ctypes.c_int64(), ctypes.c_int64()
    assert lib.byz_limits(ctypes.byref(a), ctypes.byref(b)) == 0
    assert a.value >= 1 << 20 and b.value >= 1 << 20


@pytest.mark.parametrize('n,identical', [(16385, 0), (20000, 4800)])
def test_krum_beyond_the_lds_kernels(eng, n, identical):
    f = int(0.24 * n)
    dist = point_distances(7000 + n, n, 16, identical)
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)


def sampled_picks(theta, every):
    return np.unique(np.concatenate([np.arange(0, theta, every), np.arange(min(6, theta)),
                                     np.arange(max(theta - 6, 0), theta)])).astype(np.int32)


@pytest.mark.parametrize('n,identical', [(16500, 0), (16400, 3936)])
def test_bulyan_selection_beyond_the_lds_kernels(eng, n, identical):
    """theta = n - 2f = 8580 dependent picks; every 300th and both ends are checked by the reference's own scoring pass in the state
    of that pick (scale.verify_picks), the Krum index of the full matrix completely."""
    f = int(0.24 * n)
    dist = point_distances(7100 + n, n, 12, identical)
    idx, picks = eng.krum_bulyan_select(dist, n, f)
    got = np.asarray(picks).tolist()
    theta = n - 2 * f
    assert len(got) == theta and len(set(got)) == theta
    assert idx == scale.krum_pick(dist, n, f)
    sample = sampled_picks(theta, 300)
    bad, first, expected = scale.verify_picks(dist, n, f, got, sample)
    assert bad == 0, 'pick %d: reference picks row %d, got %d (%d of %d sampled picks differ)' % (first, expected, got[first], bad, len(sample))
    print('N=%d: %d rows scored the reference\'s way over %d picks' % (n, eng.bulyan_rescored(), theta))


@pytest.mark.parametrize('n,cols,c', [(16385, 9, 7800), (20001, 6, 9600)])
def test_trimmed_mean_beyond_the_lds_kernels(eng, n, cols, c):
    rng = np.random.default_rng(7300 + n)
    g = (np.round(rng.standard_normal((n, cols)) * 256) / 256).astype(np.float32)
    assert close(eng.trimmed_mean(eng.to_device(g), n, c).numpy(), faithful.trimmed_mean(g, n, c))
    g2 = rng.standard_normal((n, cols)).astype(np.float32)
    assert close(eng.trimmed_mean(eng.to_device(g2), n, c).numpy(), faithful.trimmed_mean(g2, n, c))


@pytest.mark.parametrize('attacked', [False, True])
def test_the_defences_end_to_end_beyond_the_lds_kernels(eng, attacked):
    """Krum and Bulyan from the gradients at N = 16,640 (defences.py:23-42, :55-70): distances by the Gram kernels, sampled rows
    against fp64; Krum's index and Bulyan's aggregate against the oracle run on the ENGINE's distances (the margin between fp32 and
    fp64 distances is the subject of tests/test_gpu_scale.py, not of this file).  `attacked`: the f malicious rows are ONE vector
    (malicious.py:18-27) -- found before the Gram (dedup.hip, no row limit since this round), their distances exact zeros."""
    n, d = 16640, 96
    f = int(0.24 * n)
    rng = np.random.default_rng(99)
    g = rng.standard_normal((n, d)).astype(np.float32)
    g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if attacked:
        g[:f] = faithful.drift_vector(g[:f], 1.5)
    gd = eng.to_device(g)
    dist = eng.pairwise_distances(gd).numpy()
    if attacked:
        twins = dist[:f, :f].copy()
        np.fill_diagonal(twins, 0.0)
        assert not twins.any(), (np.count_nonzero(twins), float(np.abs(twins).max()), float(dist[0, 0]))
    rows = rng.integers(0, n, 6)
    g64 = g.astype(np.float64)
    for u in rows:
        want = np.sqrt(((g64 - g64[u]) ** 2).sum(1))
        got = dist[u].astype(np.float64).copy()
        got[u] = 0.0
        assert np.allclose(got, want, rtol=1e-5, atol=1e-5)
    ref_dist = dist.copy()
    np.fill_diagonal(ref_dist, np.inf)
    assert eng.krum(gd, n, f, return_index=True) == scale.krum_pick(ref_dist, n, f)
    out, selection = eng.bulyan(gd, n, f, return_selection=True)
    selection = np.asarray(selection.numpy()).tolist()
    sample = sampled_picks(n - 2 * f, 1500)
    bad, first, expected = scale.verify_picks(ref_dist, n, f, selection, sample)
    assert bad == 0, (first, expected)
    cols = rng.integers(0, d, 4)
    want = faithful.trimmed_mean(g[np.asarray(selection)][:, cols], n - 2 * f, 2 * f)
    assert close(out.numpy()[cols], want)


# ---- tall columns: the radix select of csrc/tall_select.hip (5,633 rows and more, the default there) --------------------------------
@pytest.mark.parametrize('n,cols,c', [(5633, 70, 1400), (6000, 64, 2880), (7001, 33, 1), (8192, 65, 4000), (10000, 130, 4800), (16384, 9, 7000)])
def test_tall_select_is_the_reference_trimmed_mean(eng, monkeypatch, n, cols, c):
    """defences.py:44-52 on columns taller than the register kernels hold: odd and even counts (one or two middle values), quarter-integer
    data (exact +t / -t ties at the window's edge: row order decides, defences.py:50), continuous data, a selection's row order, the
    extremes of the trim, a NaN, infinities; against oracle.faithful, and the sort kernels of rounds 3-6 (BYZ_TM_TALL=0) agree."""
    rng = np.random.default_rng(9000 + n)
    g = (np.round(rng.standard_normal((n, cols)) * 64) / 64).astype(np.float32)
    want = faithful.trimmed_mean(g, n, c)
    gd = eng.to_device(g)
    got = eng.trimmed_mean(gd, n, c).numpy()
    assert close(got, want)
    monkeypatch.setenv('BYZ_TM_TALL', '0')
    assert close(eng.trimmed_mean(gd, n, c).numpy(), want)
    monkeypatch.delenv('BYZ_TM_TALL')
    g2 = rng.standard_normal((n, cols)).astype(np.float32) * np.exp(rng.uniform(-3, 3, cols)).astype(np.float32)
    order = rng.permutation(n)[:max(5633, n - 300)].astype(np.int32)
    c2 = len(order) // 3
    assert close(eng.trimmed_mean(eng.to_device(g2), n, c2, row_index=order).numpy(), faithful.trimmed_mean(g2[order], len(order), c2))
    for trim in (0, n - 2):      # keep = n - 1 values (everything but the farthest) and keep = 1 (the value nearest the median)
        assert close(eng.trimmed_mean(eng.to_device(g2), n, trim).numpy(), faithful.trimmed_mean(g2, n, trim))
    g2[3, 1] = np.nan
    g2[n - 1, 2] = -np.nan
    g2[5, 4] = np.inf
    g2[6, 4] = -np.inf
    got = eng.trimmed_mean(eng.to_device(g2), n, c).numpy()
    want = faithful.trimmed_mean(g2, n, c)
    assert np.isnan(got[1]) and np.isnan(got[2]) and close(np.delete(got, [1, 2]), np.delete(want, [1, 2]))


def test_tall_select_on_a_strided_matrix_and_constant_columns(eng):
    """A leading dimension beyond the column count (a view into a wider matrix), columns that are one value (every |x - med| ties at 0),
    two values, and a column whose window edge ties across the median with both signs in a row order that matters."""
    torch = pytest.importorskip('torch')
    n, cols, c = 6400, 77, 3000
    rng = np.random.default_rng(77)
    wide = rng.standard_normal((n, cols + 19)).astype(np.float32)
    wide[:, 3] = 1.25
    wide[:, 4] = np.where(rng.random(n) < 0.5, -2.0, 2.0).astype(np.float32)
    wide[:, 5] = rng.integers(-3, 4, n).astype(np.float32)
    view = torch.from_numpy(wide).cuda()[:, :cols]
    assert view.stride(0) == cols + 19
    got = eng.trimmed_mean(view, n, c).cpu().numpy()
    assert close(got, faithful.trimmed_mean(wide[:, :cols], n, c))
