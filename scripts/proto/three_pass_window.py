"""Prototype of a three-pass median-window selection -- numpy, development aid.

What the GPU kernel would do per column, restated with the information each pass really has:

  pass 1  histogram of the column over B equal buckets of [min, max] (bucket index monotone in the value);
  scalar  from the bucket counts alone: the bucket(s) of the median rank(s), and a RANGE [a_lo, a_hi] for the left edge
          a of the kept window (the k values nearest the median are sorted positions a .. a+k-1).  The exact edge is
          the first a with s[a] + s[a+k] >= 2*med; with s[.] known only to its bucket and med only to its bucket(s),
          the earliest possible and the latest possible a bound it;
  pass 2  gather the values of the median bucket(s) and of the buckets holding sorted positions a_lo .. a_hi and
          a_lo+k-1 .. a_hi+k-1 (the only positions whose membership is undecided); sort them (<= a few dozen values);
          median exactly; threshold T = the (k - decided)-th smallest |fl(x - med)| among the undecided positions;
  pass 3  sum fl(x - med) over |.| < T, plus the first ties at == T in row order (the reference's stable sort).

Run: python scripts/proto/three_pass_window.py  -- checks bit-for-bit equality with the oracle on random, tied, duplicated
and outlier-ridden columns and reports how many values pass 2 has to gather (the capacity a kernel must provide).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import faithful  # noqa: E402

F = np.float32


def three_pass(column, keep, n_buckets=512, stats=None):
    x = np.asarray(column, dtype=F)
    rows = len(x)
    k = keep
    assert 1 <= k <= rows and np.all(np.isfinite(x))
    lo, hi = x.min(), x.max()
    if lo == hi:
        return F(F(0) + lo)   # every deviation is 0
    # ---- pass 1: histogram (fp32 bucket arithmetic as a kernel would do it; monotone in x)
    inv = F(n_buckets * (1.0 - 2.0 ** -20)) / F(hi - lo)
    bucket = np.minimum(((x - lo) * inv).astype(np.int64), n_buckets - 1)
    counts = np.bincount(bucket, minlength=n_buckets)
    cum = np.concatenate([[0], np.cumsum(counts)])          # cum[b] = values in buckets < b
    edge = lambda b: float(lo) + b / float(inv)             # noqa: E731  lower bound of bucket b (a bound, not exact)

    def bucket_of_position(p):                               # bucket holding sorted position p
        return int(np.searchsorted(cum, p, side='right') - 1)

    # ---- scalar step: median buckets and the range of the window's left edge
    r1, r2 = (rows - 1) // 2, rows // 2
    bm1, bm2 = bucket_of_position(r1), bucket_of_position(r2)
    slack = (hi - lo) * 2.0 ** -18                           # rounding of the bucket map: bounds, with margin
    mu_min, mu_max = edge(bm1) - slack, edge(bm2 + 1) + slack
    if k == rows:
        a_lo = a_hi = 0
    else:
        a_lo, a_hi = rows - k, rows - k
        for a in range(rows - k):                            # a kernel walks bucket transitions, not positions
            g_upper = edge(bucket_of_position(a) + 1) + edge(bucket_of_position(a + k) + 1) + 2 * slack
            if g_upper >= 2 * mu_min:
                a_lo = a
                break
        for a in range(a_lo, rows - k):
            g_lower = edge(bucket_of_position(a)) + edge(bucket_of_position(a + k)) - 2 * slack
            if g_lower >= 2 * mu_max:
                a_hi = a
                break
    # ---- pass 2: gather whole buckets, sort, pick sorted positions by rank
    b_left = range(bucket_of_position(a_lo), bucket_of_position(a_hi) + 1)
    b_right = range(bucket_of_position(a_lo + k - 1), bucket_of_position(a_hi + k - 1) + 1)
    wanted = sorted(set(b_left) | set(b_right) | {bm1, bm2})
    gathered = {b: np.sort(x[bucket == b]) for b in wanted}
    if stats is not None:
        stats.append(sum(len(v) for v in gathered.values()))

    def at(p):                                               # value at sorted position p (its bucket was gathered)
        b = bucket_of_position(p)
        return gathered[b][p - cum[b]]

    med = at(r1) if r1 == r2 else F(F(at(r1) + at(r2)) * F(0.5))   # np.median: mean of the two middles in fp32
    undecided = list(range(a_lo, a_hi + 1)) + list(range(max(a_lo + k - 1, a_hi + 1), a_hi + k))
    decided_inside = max(0, (a_lo + k - 1) - (a_hi + 1))     # positions a_hi+1 .. a_lo+k-2 are kept whatever a is
    need = k - decided_inside
    mags = np.sort(np.abs(np.array([at(p) for p in undecided], dtype=F) - med).astype(F))
    assert 1 <= need <= len(mags), (need, len(mags), a_lo, a_hi, k)
    thr = mags[need - 1]
    # ---- pass 3: everything strictly inside, then ties in row order (stable sort of the reference)
    dev = (x - med).astype(F)
    mag = np.abs(dev)
    inside = mag < thr
    ties = np.flatnonzero(mag == thr)[: k - int(inside.sum())]
    kept_rows = np.sort(np.concatenate([np.flatnonzero(inside), ties]))
    assert len(kept_rows) == k
    # the reference sums the kept deviations in ascending |dev| order (np.mean of the sorted slice)
    order = np.argsort(mag[kept_rows], kind='stable')
    good = dev[kept_rows][order]
    return np.mean(good) + med


def columns(rng):
    for rows, k in ((1000, 799), (1000, 1), (1000, 1000), (999, 500), (2080, 159), (5200, 399), (64, 40), (7, 3), (2, 1)):
        yield 'gaussian', rows, k, rng.standard_normal(rows).astype(F)
        yield 'quantised (many ties)', rows, k, (np.round(rng.standard_normal(rows) * 4) / 4).astype(F)
        c = rng.standard_normal(rows).astype(F)
        c[: rows // 4] = c[0]
        yield 'a quarter identical', rows, k, c
        c = rng.standard_normal(rows).astype(F)
        c[rng.integers(0, rows, size=max(1, rows // 50))] *= F(1e6)
        yield 'outliers x 1e6', rows, k, c
        yield 'heavy tails', rows, k, rng.standard_cauchy(rows).astype(F)
        c = np.repeat(rng.standard_normal((rows + 1) // 2).astype(F), 2)[:rows]
        yield 'symmetric pairs', rows, k, np.concatenate([c[: rows // 2], -c[: rows - rows // 2]]).astype(F)


def main():
    rng = np.random.default_rng(2024)
    worst = {}
    checked = 0
    for trial in range(6):
        for name, rows, k, col in columns(rng):
            for n_buckets in (256, 512):
                stats = []
                got = three_pass(col, k, n_buckets, stats)
                want = faithful.trimmed_mean_column(col, k)
                assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (name, rows, k, got, want)
                key = (name, rows, k, n_buckets)
                worst[key] = max(worst.get(key, 0), stats[0] if stats else 0)
                checked += 1
    print('%d columns: the three-pass selection equals the oracle bit for bit' % checked)
    print('values gathered in pass 2 (worst of 6 trials):')
    for (name, rows, k, n_buckets), v in sorted(worst.items(), key=lambda kv: (kv[0][1], kv[0][2], kv[0][0], kv[0][3])):
        if rows >= 999:
            print('  rows %5d keep %5d  %-22s buckets %3d : %5d' % (rows, k, name, n_buckets, v))


if __name__ == '__main__':
    main()
