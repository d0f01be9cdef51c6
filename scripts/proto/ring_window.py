"""numpy model of the ring-based three-pass median-window selection (csrc/median_window3.hip) -- development aid.

Per column of R values, B equal-width buckets of [min, max] (bucket index monotone in the value):

  pass 1  histogram; prefix sums.
  scalar  bm1, bm2 = buckets of the median ranks; ring(b) = max(bm1 - b, b - bm2, 0); N(j) = values in rings <= j;
          j* = min j with N(j) >= k;  slack s = 1 + (bm2 - bm1).
          A value of ring j deviates from the median by ((j-1) w, (j+s) w), w = bucket width, so
            rings <= j*-1-s   are among the k nearest whatever the exact median is            ("decided in"),
            rings >= j*+s+1   are not                                                          ("decided out"),
            rings j*-s..j*+s  are undecided: at most 2 (2s+1) buckets.
  pass 2  gather the median buckets and the undecided rings; ONE sort by value gives the median exactly
          (np.median: fp32 mean of the two middles); |fl(x - med)| over the sorted gathered values is V-shaped
          (bitonic), one merge gives T = the (k - N(j*-1-s))-th smallest undecided deviation.
  pass 3  sum fl(x - med) over |.| < T, ties at == T in row order; valid iff #(<T) <= k <= #(<=T) -- otherwise a
          decided-in value exceeds T (rare) and the column takes the general path.

Run: python scripts/proto/ring_window.py -- bit-for-bit against the oracle, and how many columns fall back.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import faithful  # noqa: E402

F = np.float32


class Fallback(Exception):
    pass


def ring_window(column, keep, n_buckets=512, capacity=64, stats=None):
    x = np.asarray(column, dtype=F)
    rows, k = len(x), keep
    if not (1 <= k <= rows) or not np.all(np.isfinite(x)):
        raise Fallback('k or non-finite')
    lo, hi = x.min(), x.max()
    if lo == hi:
        raise Fallback('constant column')
    inv = F(F(n_buckets) * F(1.0 - 2.0 ** -20)) / F(hi - lo)
    if not np.isfinite(inv):
        raise Fallback('range')
    bucket = np.minimum((F(x - lo) * inv).astype(np.int64), n_buckets - 1)
    counts = np.bincount(bucket, minlength=n_buckets)
    cum = np.concatenate([[0], np.cumsum(counts)])

    def bucket_of_position(p):
        return int(np.searchsorted(cum, p, side='right') - 1)

    r1, r2 = (rows - 1) // 2, rows // 2
    bm1, bm2 = bucket_of_position(r1), bucket_of_position(r2)
    s = 1 + (bm2 - bm1)
    if s > 2:
        raise Fallback('median ranks far apart')

    def n_upto(j):   # values in rings <= j
        a, b = max(bm1 - j, 0), min(bm2 + j, n_buckets - 1)
        return int(cum[b + 1] - cum[a])

    j_star = next(j for j in range(n_buckets + 1) if n_upto(j) >= k)
    j_in = j_star - 1 - s                       # rings <= j_in are decided in
    n_in = n_upto(j_in) if j_in >= 0 else 0
    ring = np.maximum(np.maximum(bm1 - bucket, bucket - bm2), 0)
    undecided = (ring > j_in) & (ring <= j_star + s)
    gathered = undecided | (ring == 0)
    if stats is not None:
        stats.append(int(gathered.sum()))
    if gathered.sum() > capacity:
        raise Fallback('capacity')
    g = np.sort(x[gathered])
    below = int(cum[max(bm1 - (j_star + s), 0)]) if True else 0
    # sorted position of gathered values: the gathered set is a union of bucket runs; position = rank among ALL values
    pos = {}
    order = np.argsort(x, kind='stable')          # (model shortcut for "count of values in lower buckets + index")
    sorted_x = x[order]
    med_lo, med_hi = sorted_x[r1], sorted_x[r2]
    # the kernel finds them in g by index arithmetic on cum[]; check that they are gathered at all
    assert gathered[order[r1]] and gathered[order[r2]]
    med = med_lo if r1 == r2 else F(F(med_lo + med_hi) * F(0.5))
    dev_g = np.abs(F(x[undecided]) - med).astype(F)
    need = k - n_in
    if not (1 <= need <= len(dev_g)):
        raise Fallback('need out of range')
    thr = np.sort(dev_g)[need - 1]
    dev = (x - med).astype(F)
    mag = np.abs(dev)
    n_lt, n_le = int((mag < thr).sum()), int((mag <= thr).sum())
    if not (n_lt <= k <= n_le):
        raise Fallback('a decided-in value exceeds the threshold')
    ties = np.flatnonzero(mag == thr)[: k - n_lt]
    kept = np.sort(np.concatenate([np.flatnonzero(mag < thr), ties]))
    good = dev[kept][np.argsort(mag[kept], kind='stable')]
    return np.mean(good) + med


def columns(rng):
    for rows, k in ((1000, 799), (1000, 1), (1000, 1000), (999, 500), (2080, 159), (5200, 399), (64, 40), (7, 3), (2, 1),
                    (100, 75), (4000, 3039), (1537, 577)):
        yield 'gaussian', rows, k, rng.standard_normal(rows).astype(F)
        yield 'scaled rows', rows, k, (rng.standard_normal(rows) * (1 + 0.5 * rng.permutation(rows) / rows)).astype(F)
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
        yield 'offset 1000', rows, k, (1000 + rng.standard_normal(rows)).astype(F)
        yield 'uniform', rows, k, rng.random(rows).astype(F)


def main():
    rng = np.random.default_rng(2025)
    results = {}
    for trial in range(8):
        for name, rows, k, col in columns(rng):
            for n_buckets, cap in ((512, 64), (1024, 128)):
                key = (name, rows, k, n_buckets, cap)
                rec = results.setdefault(key, {'ok': 0, 'fallback': {}, 'max_gather': 0})
                stats = []
                try:
                    got = ring_window(col, k, n_buckets, cap, stats)
                    want = faithful.trimmed_mean_column(col, k)
                    assert got.dtype == want.dtype and np.array_equal(got, want, equal_nan=True), (name, rows, k, got, want)
                    rec['ok'] += 1
                except Fallback as e:
                    rec['fallback'][str(e)] = rec['fallback'].get(str(e), 0) + 1
                if stats:
                    rec['max_gather'] = max(rec['max_gather'], stats[0])
    total = sum(r['ok'] for r in results.values())
    print('%d columns resolved by the ring selection, all bit-identical to the oracle' % total)
    for (name, rows, k, n_buckets, cap), r in sorted(results.items(), key=lambda kv: (kv[0][1], kv[0][2], kv[0][0], kv[0][3])):
        if rows >= 999 and (r['fallback'] or name in ('gaussian', 'scaled rows')):
            print('  rows %5d keep %5d %-22s B=%4d cap=%3d : ok %d, gathered <= %3d, fallbacks %s'
                  % (rows, k, name, n_buckets, cap, r['ok'], r['max_gather'], r['fallback'] or '-'))


if __name__ == '__main__':
    main()
