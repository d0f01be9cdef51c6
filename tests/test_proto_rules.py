"""The numpy replays of kernel-side arithmetic rules that DESIGN.md cites (scripts/proto/): they must stay true.

These are development aids, not product code: each restates, on the CPU, an exactness argument a kernel relies on (or, for
the tie rule, one that was measured on the GPU and recorded as not worth its instructions)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'scripts', 'proto', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_integer_rule_with_ties_is_the_sequential_fp32_sum():
    """csrc/select.hip (reference_score): inside one binade the left-to-right fp32 sum of a 64-entry chunk is an integer
    sum; an entry exactly half way between two multiples of the ulp adds floor + parity and leaves the sum even."""
    rule = _load('rescore_rule')
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(1, 4000))
        kind = trial % 4
        if kind == 0:
            v = np.sqrt(rng.chisquare(16, n))
        elif kind == 1:
            v = rng.integers(1, 1 << 12, n) / 64.0            # quantised: ties in almost every chunk
        elif kind == 2:
            v = np.full(n, rng.uniform(0.1, 10.0))            # one value: structural ties
        else:
            v = np.exp(rng.uniform(-20, 20, n))               # many decades: binade crossings, entries beyond 2^24 ulps
        v = np.sort(v.astype(np.float32))
        take = int(rng.integers(1, n + 1))
        v[take:] = 0.0
        want = rule.sequential(v[:take])
        got, _, _ = rule.by_rule(v)
        assert np.float32(got).view(np.uint32) == want.view(np.uint32), (trial, kind, n, take)


def _points(seed, n, dim, identical=0, quantum=None):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    if identical:
        pts[:identical] = pts[:identical].mean(axis=0)
    p = pts.astype(np.float64)
    sq = (p * p).sum(1)
    d = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (p @ p.T), 0.0)).astype(np.float32)
    d = np.minimum(d, d.T)
    if quantum:
        d = (np.round(d / quantum) * quantum).astype(np.float32)
    if identical:
        d[:identical, :identical] = 0.0
    np.fill_diagonal(d, np.inf)
    return d


def test_band_loop_selects_what_the_reference_selects():
    """csrc/large_rows.hip (more than 16,384 rows): exact fp64 scores carried through the removals, only the rows inside the
    rigorous rounding band scored the reference's way -- the selection of defences.py:59-68 pick for pick (oracle.faithful, itself
    pinned to the imported reference), on contested data, twins, exact ties, other prefix lengths, inf / negative entries."""
    from oracle import faithful
    loop = _load('band_loop')
    cases = [(_points(1, 40, 3), 40, 9), (_points(2, 57, 16), 57, 13), (_points(3, 48, 5, identical=11), 48, 11),
             (_points(4, 45, 2, quantum=0.5), 45, 10), (_points(5, 50, 4), 42, 7), (_points(6, 50, 4), 61, 12),
             (np.where(np.eye(30, dtype=bool), np.inf, 3.0).astype(np.float32), 30, 6)]
    odd = _points(7, 44, 3)
    odd[3, 9] = odd[9, 3] = np.inf
    odd[5, 20] = odd[20, 5] = -1e-3
    odd[7, 8] = odd[8, 7] = np.inf
    cases.append((odd, 44, 10))
    scored = 0
    for dist, users, corrupted in cases:
        stats = {}
        assert loop.selection(dist, users, corrupted, stats) == faithful.bulyan_selection(dist, users, corrupted)
        scored += stats['scored']
    assert scored > 0
    huge = np.where(np.eye(30, dtype=bool), np.inf, 1e19).astype(np.float32)       # every score >= 1e20: KeyError(-1), defences.py:65
    for fn in (loop.selection, faithful.bulyan_selection):
        try:
            fn(huge, 30, 6)
            raise AssertionError('no KeyError')
        except KeyError:
            pass


def test_band_holds_the_sequential_fp32_sum():
    """The band itself: fl(sum) of m non-negative fp32 terms added left to right lies within (1 -+ u)^(m-1) of the exact sum, and
    (1 + 2.1 m u) covers ((1 + u) / (1 - u))^(m-1) up to m = 2^18 (the kernel scores every live row beyond that)."""
    loop = _load('band_loop')
    rng = np.random.default_rng(3)
    for trial in range(40):
        m = int(rng.integers(2, 30000))
        kind = trial % 4
        v = (np.sqrt(rng.chisquare(8, m)) if kind == 0 else np.exp(rng.uniform(-12, 12, m)) if kind == 1
             else rng.integers(0, 1 << 14, m) / 128.0 if kind == 2 else np.full(m, rng.uniform(0.1, 5.0)))
        v = np.sort(v.astype(np.float32))
        exact = float(v.astype(np.float64).sum())
        got = float(np.cumsum(v, dtype=np.float32)[-1])          # (numpy's cumsum IS the left-to-right chain)
        assert exact * (1.0 - loop.U) ** (m - 1) <= got <= exact * (1.0 + loop.U) ** (m - 1)
    for m in (2, 100, 16384, 65536, 1 << 18):
        assert ((1.0 + loop.U) / (1.0 - loop.U)) ** (m - 1) <= loop.band_factor(m)
