"""How the Bulyan loop's band of contenders moves from pick to pick -- the design study behind DESIGN.md section 7, item 3.

The loop (csrc/select.hip) decides every pick in the reference's own arithmetic: rows whose EXACT score (fp64) lies within
2.2 delta of the minimum (delta = u (m + 1) / 2, the bound of a sequential fp32 sum of m terms) are re-scored as the reference
would (one ~20 us chain each, in parallel) and the smallest fp32 score wins.  An incremental re-score
(scripts/proto/seqsum_incr.py) makes a row that was scored at the previous pick cheap (~3-5 us); this script asks what is left
on the critical path then.  It replays the loop on the `scaled` family and follows a TRACKED set:

  * a row is admitted when its exact score comes within `admit` x delta of the minimum; its first full re-score runs in the
    background and is ready `lead` picks later (an idle wave, a snapshot of the row; the picks made meanwhile are caught up
    incrementally); from then on it is updated at every pick;
  * at a pick, a contender that is not ready yet can be left out if it cannot win: its score is at least E (1 - delta), and if
    that exceeds the best fp32 score among the ready contenders the pick is decided without it; otherwise the pick STALLS on a
    full re-score (the critical path of today's loop at every pick that has a contender).

`band_scale` widens delta so that a 2000-row replay has the contender density of N = 10,000 (delta grows with m, the spacing of
the scores shrinks with N: ~25x between N = 2000 and N = 10,000).

    python scripts/proto/band_dynamics.py [N] [band_scale]
"""
import sys

import numpy as np

U = 2.0 ** -24


def scaled_family(n, d, seed):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    s = (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)
    return g * s[:, None]


def distances(g):
    g64 = g.astype(np.float64)
    sq = (g64 * g64).sum(axis=1)
    d2 = np.maximum(sq[:, None] + sq[None, :] - 2.0 * g64 @ g64.T, 0.0)
    d = np.sqrt(d2).astype(np.float32)
    np.fill_diagonal(d, 0.0)
    return d


def replay(n, band_scale, admit_levels=(2.2, 3.0, 4.0, 6.0, 8.0), leads=(2, 4, 6), d=512, seed=5):
    f = int(0.24 * n)
    theta = n - 2 * f
    dist = distances(scaled_family(n, d, seed))
    big = np.float32(np.inf)
    present = np.ones(n, dtype=bool)
    schemes = {(a, l): {'since': {}, 'stalls': 0, 'unready': 0, 'background': 0, 'tracked_sum': 0} for a in admit_levels for l in leads}
    last_in_band = {}
    reentries = 0
    contenders_sum = entrants_sum = today_with_rescore = 0
    prev_band = set()
    for pick in range(theta):
        idx = np.flatnonzero(present)
        m = len(idx) - f                       # defences.py:33: the n' - f smallest of the n' - 1 others
        sub = dist[np.ix_(idx, idx)].astype(np.float64)
        np.fill_diagonal(sub, np.inf)
        part = np.partition(sub, m - 1, axis=1)[:, :m]
        exact = part.sum(axis=1)
        delta = U * (m + 1) / 2 * band_scale
        e_min = exact.min()
        in_band = np.flatnonzero(exact <= e_min * (1.0 + 2.2 * delta))
        # the reference's decision: sequential fp32 sums of the contenders, smallest (earliest) wins
        s32 = {}
        for j in in_band:
            row = np.sort(part[j]).astype(np.float32)
            s32[j] = float(np.cumsum(row, dtype=np.float32)[-1])
        winner_local = min(in_band, key=lambda j: (s32[j], idx[j]))
        band_rows = set(int(idx[j]) for j in in_band)
        contenders_sum += len(in_band)
        entrants_sum += len(band_rows - prev_band)
        reentries += sum(1 for r in band_rows - prev_band if pick - last_in_band.get(r, -10**9) <= 8)
        for r in band_rows:
            last_in_band[r] = pick
        today_with_rescore += 1 if len(in_band) > 1 else 0
        prev_band = band_rows
        for (admit, lead), st in schemes.items():
            near = np.flatnonzero(exact <= e_min * (1.0 + admit * delta))
            since = st['since']
            for j in near:
                r = int(idx[j])
                if r not in since:
                    since[r] = pick
                    st['background'] += 1
            for r in [r for r in since if not present[r]]:
                del since[r]
            ready = [j for j in in_band if pick - since.get(int(idx[j]), pick) >= lead]
            waiting = [j for j in in_band if j not in ready]
            if len(in_band) > 1 and waiting:
                st['unready'] += 1
                best_ready = min((s32[j] for j in ready), default=np.inf)
                if any(exact[j] * (1.0 - delta) <= best_ready for j in waiting):
                    st['stalls'] += 1
            st['tracked_sum'] += len(since)
        present[idx[winner_local]] = False
    print('N = %d (theta = %d picks), band widened %gx: %.1f contenders and %.2f entrants per pick; today %d of %d picks carry a '
          'full re-score on their critical path; %d of the %d entrants had been in the band within the last 8 picks'
          % (n, theta, band_scale, contenders_sum / theta, entrants_sum / theta, today_with_rescore, theta, reentries, entrants_sum))
    for (admit, lead), st in sorted(schemes.items()):
        print('  admit at %.1f delta, ready after %d picks: %4d of %d picks stall on a full re-score (%.1f%%; %d = %.1f%% without the '
              'lower-bound rule: some contender is not ready); %.1f rows tracked per pick, %d background re-scores in all'
              % (admit, lead, st['stalls'], theta, 100.0 * st['stalls'] / theta, st['unready'], 100.0 * st['unready'] / theta,
                 st['tracked_sum'] / theta, st['background']))


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    scale = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
    replay(n, scale)
