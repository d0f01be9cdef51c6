"""The Bulyan loop of csrc/large_rows.hip (more than 16,384 rows) replayed in numpy: exact fp64 scores carried through the
removals, the rigorous rounding band, the contenders scored the reference's way.  A development aid, not product code;
tests/test_proto_rules.py runs it against the oracle on the CPU, where the kernel itself cannot run.

The claim the kernel rests on: with u = 2^-24 the left-to-right fp32 sum of m non-negative terms lies within
(1 -+ u)^(m-1) of their exact sum s, so a row can reach the smallest fp32 sum only if its exact sum is at most
s_min ((1 + u) / (1 - u))^(m-1) <= s_min (1 + 2.1 m u)   (m u <= 2^-6).  Rows outside that band are never scored, rows inside
are scored exactly as defences.py:33-34 does, and the reference's strict '<' in visit order 1, 0, 2, ... decides.
"""
import numpy as np

U = 2.0 ** -24
KRUM_INIT = np.float32(1e20)


def visit_position(u):
    return 1 if u == 0 else (0 if u == 1 else u)


def sequential_f32(values):
    s = np.float32(0.0)
    for v in values:
        s = np.float32(s + np.float32(v))
    return s


def band_factor(m):
    return 1.0 + 2.1 * m * U + 1e-9


def selection(dist, users_count, corrupted, stats=None):
    """defences.py:59-68 on a dense fp32 matrix (diagonal ignored): the indices in selection order; KeyError where the
    reference raises it.  `stats` (a dict) receives the number of rows scored the reference's way."""
    d = np.asarray(dist, dtype=np.float32)
    n = d.shape[0]
    theta = users_count - 2 * corrupted
    order = np.empty((n, n - 1), dtype=np.int64)
    vals = np.empty((n, n - 1), dtype=np.float32)
    for u in range(n):
        others = np.array([c for c in range(n) if c != u], dtype=np.int64)
        row = d[u, others]
        # NaN behind +inf, either sign (select.hip's keys); a stable sort keeps equal values in column order, as the keys do
        key = np.where(np.isnan(row), np.float32(np.inf), row)
        tie = np.isnan(row).astype(np.int64)
        idx = np.lexsort((others, tie, key))
        order[u], vals[u] = others[idx], row[idx]
    irregular = np.array([bool(np.any(~(vals[u] >= 0)) or np.any(~np.isfinite(vals[u]))) for u in range(n)])
    drop = max(0, min(n - 1, (n - 1) - users_count + corrupted))
    total = np.array([vals[u].astype(np.float64).sum() if not irregular[u] else 0.0 for u in range(n)])
    top = np.array([vals[u][n - 1 - drop:].astype(np.float64).sum() if (drop and not irregular[u]) else 0.0 for u in range(n)])
    top_first = np.full(n, n - 1 - drop, dtype=np.int64)
    rank = np.empty((n, n), dtype=np.int64)
    for u in range(n):
        rank[u, order[u]] = np.arange(n - 1)
    scale = float(total.max()) if n else 0.0
    gone = np.zeros(n, dtype=bool)
    picked, scored = [], 0
    for t in range(theta):
        want, left = users_count - t - corrupted, n - t - 1
        m = left if want >= left else max(want, 0)
        all_of_them = want >= left
        live = ~gone
        regular = live & ~irregular
        exact = np.where(all_of_them, total, total - top)
        low = exact[regular].min() if regular.any() else np.inf
        bound = low * band_factor(m) + scale * 2.0 ** -36
        if not low < 9e19:
            bound = np.inf
        contenders = [u for u in range(n) if live[u] and (irregular[u] or exact[u] <= bound)]
        scored += len(contenders)
        best, best_pos, winner = KRUM_INIT, None, -1
        for u in contenders:
            walk = [vals[u][r] for r in range(n - 1) if not gone[order[u][r]]][:m]
            s = sequential_f32(walk)
            if s < best or (s == best and winner >= 0 and visit_position(u) < best_pos):
                best, best_pos, winner = s, visit_position(u), u
        if winner < 0:
            raise KeyError(-1)
        picked.append(winner)
        gone[winner] = True
        for u in range(n):
            if gone[u] or irregular[u]:
                continue
            r = rank[u, winner]
            v = float(vals[u][r])
            total[u] -= v
            if drop and r >= top_first[u]:
                p = top_first[u] - 1
                while p >= 0 and gone[order[u][p]]:
                    p -= 1
                top[u] += (float(vals[u][p]) if p >= 0 else 0.0) - v
                top_first[u] = p
    if stats is not None:
        stats['scored'] = scored
    return picked
