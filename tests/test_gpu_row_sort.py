"""The row sort of the selection kernels (csrc/select.hip, reference defences.py:33-34: `sorted(distances[user].values())`): the
register-blocked bitonic network (round 6: four levels per pass over LDS, 32 passes instead of 105 at 16,384 keys) against the
textbook form (BYZ_ROW_SORT_BLOCKED=0) -- the order of the 64-bit keys (value, column) is unique, so everything derived from
it must be the same bits: every row's Krum score (the sequential fp32 sum of its sorted prefix), Krum's index and Bulyan's
selection pick for pick -- and against the C oracle.  Inputs with what a sort can get wrong: exact ties, zeros (identical
clients), +inf and NaN entries, every padded size from 256 to 16,384 keys and row counts off the powers of two."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def distances(n, seed, ties=True, special=True):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, 8)).astype(np.float32)
    if ties:
        pts[n // 3: n // 3 + 5] = pts[n // 3]              # identical clients: exact zeros, identical rows
        pts = np.round(pts * 4) / 4 if n <= 600 else pts      # a lattice: many exactly equal distances
    d = np.sqrt(((pts[:, None, :].astype(np.float64) - pts[None, :, :]) ** 2).sum(-1)).astype(np.float32) if n <= 3000 else None
    if d is None:
        g = pts.astype(np.float64)
        sq = (g * g).sum(1)
        d = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * g @ g.T, 0.0)).astype(np.float32)
        d = np.maximum(d, d.T)
    if special and n >= 40:
        d[7, :] = d[:, 7] = np.inf                            # a client with an infinite gradient
        d[11, 5] = d[5, 11] = np.nan                          # and a NaN pair
    np.fill_diagonal(d, np.inf)
    return np.ascontiguousarray(d)


def select_both_ways(eng, dist, n, f):
    from attacking_federate_learning_amd.engine import _check, _vp
    out = {}
    dev = eng.to_device(dist)
    for mode in ('0', '1'):
        os.environ['BYZ_ROW_SORT_BLOCKED'] = mode
        try:
            scores = eng.empty((n,), np.float32)
            idx = ctypes.c_int32(-2)
            _check(eng.lib.byz_krum_select_dev(eng.ctx, _vp(dev.ptr), n, n, f, ctypes.byref(idx), _vp(scores.ptr), None))
            try:
                sel = list(eng.bulyan_select(dist, n, f))
            except KeyError:
                sel = 'KeyError'
            out[mode] = (int(idx.value), scores.numpy().view(np.uint32).copy(), sel)
        finally:
            os.environ.pop('BYZ_ROW_SORT_BLOCKED', None)
    return out


@pytest.mark.parametrize('n', [255, 256, 257, 300, 511, 513, 1000, 1025, 2049, 3000, 4097, 5000])
def test_blocked_network_is_the_textbook_network_bit_for_bit(eng, n):
    f = int(0.24 * n)
    dist = distances(n, 900 + n)
    got = select_both_ways(eng, dist, n, f)
    assert got['0'][0] == got['1'][0]
    assert np.array_equal(got['0'][1], got['1'][1])           # every row's score, as bits (NaN scores included)
    assert got['0'][2] == got['1'][2]


@pytest.mark.parametrize('n', [8200, 10000])
def test_blocked_network_at_the_largest_sizes(eng, n):
    """configs[4]'s row count and the first size of the 16,384-key network; clean data (the C oracle's replay is the check)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import scale
    f = int(0.24 * n)
    dist = distances(n, 77 + n, ties=False, special=False)
    got = select_both_ways(eng, dist, n, f)
    assert got['0'][0] == got['1'][0] == scale.krum_pick(dist, n, f)
    assert np.array_equal(got['0'][1], got['1'][1])
    assert got['0'][2] == got['1'][2]
