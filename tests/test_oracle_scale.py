"""The C restatement of the selection loops (oracle/scale.py) against the numpy oracle and the golden vectors.

oracle.faithful is pinned bit for bit against the imported reference (tests/test_oracle_vs_reference.py) and the
golden vectors; the C oracle exists only because faithful.bulyan_selection needs hours at N = 4000.  Here the two
must agree wherever both run, in both arithmetics."""
import numpy as np
import pytest

from oracle import faithful, ideal, scale


def point_distances(seed, n, dim=12):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    dist = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)).astype(np.float32)
    np.fill_diagonal(dist, np.inf)
    return dist


@pytest.mark.parametrize('case', ['krum_iid_10x257', 'krum_scaled_33x1000', 'krum_attacked_12x300',
                                  'krum_allsame_6x64', 'krum_f0_5x40'])
def test_krum_pick_on_the_reference_distances(golden, case):
    c = golden[case]
    n, f = len(c['G']), int(c['f'])
    assert scale.krum_pick(c['dist'], n, f) == int(c['index'])


@pytest.mark.parametrize('case', ['bulyan_iid_11x200', 'bulyan_boundary_15x120', 'bulyan_scaled_40x500',
                                  'bulyan_attacked_23x150', 'bulyan_f0_6x30'])
def test_bulyan_selection_on_the_golden_cases(golden, case):
    c = golden[case]
    g, f, n = c['G'], int(c['f']), len(c['G'])
    dist = faithful.distance_matrix(g)
    assert scale.bulyan_selection(dist, n, f) == c['selection'].tolist()


@pytest.mark.parametrize('n,f', [(2, 0), (3, 0), (7, 1), (23, 5), (64, 15), (130, 31), (257, 60)])
def test_equals_the_numpy_oracle_in_both_arithmetics(n, f):
    dist = point_distances(100 + n, n)
    assert scale.krum_pick(dist, n, f) == faithful.krum_pick(dist, faithful.visit_order(n), n, f)
    assert scale.krum_pick(dist, n, f, mode='ideal') == ideal.krum_index(dist, n, f)
    if n >= 4 * f + 3:
        assert scale.bulyan_selection(dist, n, f) == faithful.bulyan_selection(dist, n, f)
        got, margins = scale.bulyan_selection(dist, n, f, mode='ideal', with_margins=True)
        want, want_margins = ideal.bulyan_selection(dist, n, f, with_margins=True)
        assert got == want
        assert np.allclose(margins[:-1], want_margins[:-1], rtol=1e-6, atol=1e-12)


def test_scores_are_the_sequential_fp32_sums():
    n, f = 90, 20
    dist = point_distances(7, n)
    idx, _, scores = scale.krum_pick(dist, n, f, with_scores=True)
    want = faithful.krum_scores(dist, faithful.visit_order(n), n, f)
    assert all(np.float32(scores[u]) == want[u] for u in range(n))
    assert idx == faithful.krum_pick(dist, faithful.visit_order(n), n, f)


def test_identical_rows_tie_and_visit_order_decides():
    n, f = 41, 9
    dist = point_distances(11, n)
    dist[:f, :] = dist[0, :]
    dist[:, :f] = dist[:, [0]]
    dist[:f, :f] = 0.0
    np.fill_diagonal(dist, np.inf)
    assert scale.bulyan_selection(dist, n, f) == faithful.bulyan_selection(dist, n, f)
    assert scale.krum_pick(dist, n, f) == faithful.krum_pick(dist, faithful.visit_order(n), n, f)


def test_users_count_other_than_the_row_count_follows_python_slices():
    # krum(..., users_count) with users_count != len(G): the prefix length is a plain Python slice (defences.py:34)
    n = 30
    dist = point_distances(5, n)
    for users_count, f in ((40, 3), (20, 4), (3, 5)):
        assert scale.krum_pick(dist, users_count, f) == faithful.krum_pick(dist, faithful.visit_order(n), users_count, f)


def test_no_winner_raises_like_the_reference():
    dist = np.full((5, 5), 1e30, dtype=np.float32)     # every score overflows 1e20
    np.fill_diagonal(dist, np.inf)
    with pytest.raises(KeyError):
        scale.bulyan_selection(dist, 5, 0)
    with pytest.raises(KeyError):
        faithful.bulyan_selection(dist, 5, 0)


def test_verify_picks_agrees_with_the_full_selection():
    n, f = 120, 28
    dist = point_distances(21, n)
    sel = scale.bulyan_selection(dist, n, f)
    assert scale.verify_picks(dist, n, f, sel, np.arange(len(sel))) == (0, -1, -1)
    wrong = list(sel)
    wrong[30], wrong[31] = wrong[31], wrong[30]
    bad, first, expected = scale.verify_picks(dist, n, f, wrong, np.arange(len(sel)))
    assert bad >= 1 and first == 30 and expected == sel[30]


def test_replay_selection_scores_a_selection_in_its_own_state():
    """SURVEY.md 8(d), the margin protocol's second clause: a selection made on other numbers is replayed pick by pick with
    ITS OWN picks removed.  Replaying the rule's own selection gives zero excess, the rule's own margins and argmin == pick;
    a selection that parts ways at one pick is charged exactly the score ratio of that pick, and what follows is judged in
    the state that pick left behind (not the oracle's)."""
    n, f = 60, 14
    dist = point_distances(31, n)
    want, margins = scale.bulyan_selection(dist, n, f, mode='ideal', with_margins=True)
    excess, margin, argmin = scale.replay_selection(dist, n, f, want, mode='ideal')
    assert np.all(excess == 0.0) and argmin.tolist() == want and np.array_equal(margin, margins)
    # part ways at pick 5: take the runner-up instead of the winner
    _, _, scores = scale.krum_pick(dist, n, f, mode='ideal', with_scores=True)
    removed = np.zeros(n, dtype=bool)
    removed[want[:5]] = True
    live = np.flatnonzero(~removed)
    sub = dist[np.ix_(live, live)]
    # the rule's scores in that state, by the numpy oracle: prefix length (n - 5) - f of the live rows' sorted lists
    take = (n - 5) - f
    sc = np.array([np.sort(np.delete(sub[i], i).astype(np.float64))[:take].sum() for i in range(len(live))])
    order = np.argsort(sc, kind='stable')
    assert live[order[0]] == want[5]
    runner_up = int(live[order[1]])
    mine = want[:5] + [runner_up]
    # continue with the rule itself from the state the runner-up left behind
    gone = set(mine)
    while len(mine) < n - 2 * f:
        live = np.array([r for r in range(n) if r not in gone])
        sub = dist[np.ix_(live, live)]
        take = max(min(n - len(mine) - f, len(live) - 1), 0)
        sc = np.array([np.sort(np.delete(sub[i], i).astype(np.float64))[:take].sum() for i in range(len(live))])
        # visit order 1, 0, 2, ... with a strict '<' (defences.py:27-37)
        visit = sorted(range(len(live)), key=lambda k: (0 if live[k] == 1 else 1 if live[k] == 0 else 2, live[k]))
        best, best_k = 1e20, -1
        for k in visit:
            if sc[k] < best:
                best, best_k = sc[k], k
        mine.append(int(live[best_k]))
        gone.add(int(live[best_k]))
    excess, margin, argmin = scale.replay_selection(dist, n, f, mine, mode='ideal')
    want_excess = (np.sort(sc_at_pick5(dist, want, n, f))[1] / np.sort(sc_at_pick5(dist, want, n, f))[0]) - 1.0
    assert excess[5] == pytest.approx(want_excess, rel=1e-12) and excess[5] > 0.0
    assert np.all(np.delete(excess, 5) == 0.0)                 # every other pick is the argmin of the state it was made in
    assert argmin[5] == want[5] and np.array_equal(np.delete(argmin, 5), np.delete(np.array(mine, dtype=np.int32), 5))
    with pytest.raises(ValueError):
        scale.replay_selection(dist, n, f, want[:3] + want[:3], mode='ideal')


def sc_at_pick5(dist, want, n, f):
    removed = np.zeros(n, dtype=bool)
    removed[want[:5]] = True
    live = np.flatnonzero(~removed)
    sub = dist[np.ix_(live, live)]
    take = (n - 5) - f
    return np.array([np.sort(np.delete(sub[i], i).astype(np.float64))[:take].sum() for i in range(len(live))])
