"""The speculative Bulyan loop (round 6, csrc/select.hip bulyan_spec_kernel; needs an MI355X).

Batches of picks are decided on the rows' exact scores and the contested ones verified behind the batch, each in the state of its
pick; a batch whose optimistic winner was not the reference's at some pick is rolled back to that pick.  Whatever the batch length,
the selection must be the loop's of rounds 2-5 (`BYZ_BULYAN_BATCH=0`: every contested pick re-scored before the next one) and the
reference's own (oracle/scale.py: defences.py:59-68 restated in C, its arithmetic), pick for pick -- on data that contests most
picks, on exact ties, twins, infinite and negative entries, on matrices where the reference gives up (KeyError) at the first pick
and in the middle of the loop.  Below 1000 rows the default is the sequential loop, so every case here FORCES the batches.
"""
import numpy as np
import pytest

from oracle import scale

pytestmark = pytest.mark.gpu

BATCHES = ('1', '3', '24', '32')


def point_distances(seed, n, dim, identical=0, quantum=None):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    pts *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if identical:
        pts[:identical] = pts[:identical].mean(axis=0)
    p64 = pts.astype(np.float64)
    sq = (p64 * p64).sum(1)
    d = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (p64 @ p64.T), 0.0)).astype(np.float32)
    d = np.minimum(d, d.T)
    if quantum:   # distances on a coarse grid: exact ties between rows that are not twins, sums that are exact in fp32
        d = (np.round(d / quantum) * quantum).astype(np.float32)
    if identical:
        d[:identical, :identical] = 0.0
        d[:identical, :] = d[0, :]
        d[:, :identical] = d[:, [0]]
    np.fill_diagonal(d, np.inf)
    return d


def selections(eng, monkeypatch, dist, users, corrupted, batches=BATCHES):
    """{batch setting: selection or the exception's type}"""
    out = {}
    for b in ('0',) + tuple(batches):
        monkeypatch.setenv('BYZ_BULYAN_BATCH', b)
        try:
            out[b] = np.asarray(eng.bulyan_select(dist, users, corrupted)).tolist()
        except KeyError:
            out[b] = KeyError
    monkeypatch.delenv('BYZ_BULYAN_BATCH')
    return out


def oracle_selection(dist, users, corrupted):
    try:
        return scale.bulyan_selection(dist, users, corrupted)
    except KeyError:
        return KeyError


@pytest.mark.parametrize('n,dim,identical,quantum', [
    (300, 4, 0, None),          # few dimensions: central rows crowd, nearly every pick contested
    (777, 16, 0, None),
    (1100, 16, 0, None),        # five workgroups, the last one ragged
    (1500, 2000, 0, None),      # the bench's kind of data: scores a rounding error apart
    (900, 16, 216, None),       # the attack's twins: one class many rows wide
    (640, 300, 77, None),
    (520, 8, 0, 0.25),          # exact ties between rows that are not twins: the visit order decides, in every state
    (1300, 3, 0, 0.5),
])
def test_batches_select_what_the_sequential_loop_selects(eng, monkeypatch, n, dim, identical, quantum):
    dist = point_distances(4200 + n, n, dim, identical, quantum)
    f = int(0.24 * n)
    got = selections(eng, monkeypatch, dist, n, f)
    want = oracle_selection(dist, n, f)
    assert got['0'] == want
    for b in BATCHES:
        assert got[b] == want, (b, next(i for i, (x, y) in enumerate(zip(got[b], want)) if x != y))
    # which workgroup owns which row (dealt in the order of the first scores, the twins' copies in front: the default; row 256 g + i
    # to thread i of workgroup g: BYZ_BULYAN_DEAL=0) changes who scores a pair, not the selection
    monkeypatch.setenv('BYZ_BULYAN_DEAL', '0')
    assert selections(eng, monkeypatch, dist, n, f, batches=('24',))['24'] == want
    monkeypatch.delenv('BYZ_BULYAN_DEAL')
    eng.check()


def test_batches_on_a_matrix_that_contests_everything(eng, monkeypatch):
    """All distances equal: every row is in the band at every pick (one (pick, row) pair per row and pick of a batch: the lists'
    capacity), every sum is exact, and the visit order 1, 0, 2, 3, ... decides every pick."""
    n, f = 600, 100
    dist = np.full((n, n), 3.0, dtype=np.float32)
    np.fill_diagonal(dist, np.inf)
    got = selections(eng, monkeypatch, dist, n, f, batches=('5', '32'))
    want = oracle_selection(dist, n, f)
    assert want[:4] == [1, 0, 2, 3]
    assert got['0'] == want and got['5'] == want and got['32'] == want


@pytest.mark.parametrize('users_delta,corrupted', [(0, 60), (0, 1), (-40, 50), (25, 30)])
def test_batches_with_other_prefix_lengths(eng, monkeypatch, users_delta, corrupted):
    """users_count need not be the row count (defences.py:59-68 is called with whatever the caller passes): the prefix of a pick
    is users_count - t - f of the n - t - 1 live entries, clamped both ways."""
    n = 420
    dist = point_distances(77, n, 6)
    users = n + users_delta
    got = selections(eng, monkeypatch, dist, users, corrupted, batches=('4', '32'))
    want = oracle_selection(dist, users, corrupted)
    assert got['0'] == want and got['4'] == want and got['32'] == want


def test_batches_with_non_finite_and_negative_entries(eng, monkeypatch):
    """inf inside a row's prefix keeps the row out of a pick (its score is not below 1e20; NaN: the order Python's sort leaves is
    not specified); a NEGATIVE entry (a caller's
    matrix need not be a metric) sends that row's re-score to the literal chain, whose liveness comes from a bitmap of the
    columns in the state of ITS pick."""
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
    got = selections(eng, monkeypatch, dist, n, f, batches=('1', '6', '32'))
    want = oracle_selection(dist, n, f)
    assert got['0'] == want
    for b in ('1', '6', '32'):
        assert got[b] == want, b


def test_the_reference_gives_up_at_the_first_pick_and_in_the_middle(eng, monkeypatch):
    """No score below 1e20: the reference pops key -1 (KeyError, defences.py:65).  At the first pick (every distance 1e19: a sum
    of 100 of them is 1e21), and in the MIDDLE of the loop: a row's score never grows from pick to pick, so that only happens when
    the rows that can be picked run out -- fifty ordinary rows, and 350 that are infinitely far from each other (more non-finite
    entries than a prefix leaves out): pick 50 finds nobody, in the middle of a batch whose earlier picks were contested."""
    n, f = 400, 40
    huge = np.full((n, n), 1e19, dtype=np.float32)
    np.fill_diagonal(huge, np.inf)
    got = selections(eng, monkeypatch, huge, n, f, batches=('7', '32'))
    assert oracle_selection(huge, n, f) is KeyError
    assert got['0'] is KeyError and got['7'] is KeyError and got['32'] is KeyError
    mid = point_distances(15, n, 3)
    mid[50:, 50:] = np.inf
    assert oracle_selection(mid, n, f) is KeyError
    got = selections(eng, monkeypatch, mid, n, f, batches=('7', '32'))
    assert got['0'] is KeyError and got['7'] is KeyError and got['32'] is KeyError
    # ... and with a theta the fifty rows can fill, the same matrix selects them, in the reference's order
    users, corrupted = 130, 40
    want = oracle_selection(mid, users, corrupted)
    got = selections(eng, monkeypatch, mid, users, corrupted, batches=('7', '32'))
    assert want is not KeyError and sorted(want) == list(range(50))
    assert got['0'] == want and got['7'] == want and got['32'] == want


@pytest.mark.parametrize('n', [4000])
def test_batches_at_the_headline_size(eng, monkeypatch, n):
    """N = 4000 (sixteen workgroups, theta = 2080) on distances that contest three picks in four: the default batches against
    the sequential loop; a dozen picks of the loop are not the exact minimum's (roll-backs happen)."""
    dist = point_distances(n, n, 4096 if n <= 4000 else 512)
    f = int(0.24 * n)
    monkeypatch.setenv('BYZ_BULYAN_BATCH', '0')
    want = np.asarray(eng.bulyan_select(dist, n, f)).tolist()
    rescored = eng.bulyan_rescored()
    monkeypatch.delenv('BYZ_BULYAN_BATCH')
    got = np.asarray(eng.bulyan_select(dist, n, f)).tolist()
    assert got == want
    assert rescored > 1000 and eng.bulyan_rescored() >= rescored    # (discarded picks' re-scores are counted too)
    assert len(set(got)) == n - 2 * f
