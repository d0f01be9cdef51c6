"""fp64 evaluation of the same selection rules.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Where ``oracle.faithful`` reproduces the reference's fp32 rounding, this module
computes what the reference *approximates*: exact-to-fp64 distances and scores,
vectorised so it finishes in seconds at sizes the reference cannot run.  It also
reports decision margins, which is what the parity protocol of SURVEY.md section
8(d) needs: an index must match wherever the margin exceeds the fp32 noise of
the reference's own arithmetic; inside the noise any candidate within the margin
is as good as the reference's.

Rules kept from the reference: visit order 1, 0, 2, ...; strict ``<``; unsquared
norms; prefix length ``users_count - corrupted_count``; median-window selection
with row-order tie-break; selection-order rows into Bulyan's trimmed mean.
"""
import numpy as np

from .faithful import python_prefix_len, visit_order


def distance_matrix(users_grads, chunk=1 << 16):
    """Unsquared pairwise L2 distances in fp64 via the Gram identity; diagonal = +inf.

    Identical rows give exactly 0 because every term comes from the same fp64 Gram.
    """
    g = np.asarray(users_grads)
    n, d = g.shape
    gram = np.zeros((n, n), dtype=np.float64)
    for s in range(0, d, chunk):
        blk = g[:, s:s + chunk].astype(np.float64)
        gram += blk @ blk.T
    sq = np.diag(gram)
    d2 = sq[:, None] + sq[None, :] - 2.0 * gram
    # exact zeros for bitwise-identical rows even if the dgemm is not perfectly symmetric
    d2 = np.minimum(d2, d2.T)
    dist = np.sqrt(np.maximum(d2, 0.0))
    np.fill_diagonal(dist, np.inf)
    return dist


def krum_scores(dist, alive_mask, users_count, corrupted_count):
    """fp64 score per row (inf for removed rows): sum of the smallest ``keep`` live distances."""
    alive = np.flatnonzero(alive_mask)
    sub = dist[np.ix_(alive, alive)].astype(np.float64)
    sub_sorted = np.sort(sub, axis=1)[:, :-1]  # the +inf self entry sorts last
    cnt = python_prefix_len(sub_sorted.shape[1], users_count - corrupted_count)
    scores = np.full(dist.shape[0], np.inf)
    scores[alive] = sub_sorted[:, :cnt].sum(axis=1) if cnt else 0.0
    return scores


def pick(scores, alive_mask):
    """argmin in the reference's visit order with strict '<' (ties -> earlier in 1,0,2,...)."""
    best, best_idx = 1e20, -1
    for u in visit_order(len(scores)):
        if alive_mask[u] and scores[u] < best:
            best, best_idx = scores[u], u
    return best_idx


def margin(scores, alive_mask, winner):
    """Relative gap between the winner's score and the runner-up's."""
    live = np.flatnonzero(alive_mask)
    others = scores[live[live != winner]]
    if others.size == 0:
        return np.inf
    gap = others.min() - scores[winner]
    return gap / max(abs(scores[winner]), np.finfo(np.float64).tiny)


def krum_index(dist, users_count, corrupted_count, with_margin=False):
    alive = np.ones(dist.shape[0], dtype=bool)
    scores = krum_scores(dist, alive, users_count, corrupted_count)
    idx = pick(scores, alive)
    if with_margin:
        return idx, margin(scores, alive, idx), scores
    return idx


def bulyan_selection(dist, users_count, corrupted_count, with_margins=False):
    """Selection order of Bulyan's loop with fp64 scores (and the margin of every pick)."""
    n = dist.shape[0]
    set_size = users_count - 2 * corrupted_count
    alive = np.ones(n, dtype=bool)
    picked, margins = [], []
    dist64 = dist.astype(np.float64)
    for t in range(set_size):
        scores = krum_scores(dist64, alive, users_count - t, corrupted_count)
        idx = pick(scores, alive)
        picked.append(idx)
        if with_margins:
            margins.append(margin(scores, alive, idx))
        alive[idx] = False
    if with_margins:
        return picked, np.asarray(margins)
    return picked


def trimmed_mean(rows, corrupted_count, block=4096):
    """defences.py:44-52 with the fp32 median/deviation rule and an fp64 mean.

    The kept *set* follows the reference exactly (fp32 median, fp32 deviations,
    stable |dev| order); only the final mean is accumulated in fp64, which moves
    the result by a few fp32 ulps at most.
    """
    g = np.asarray(rows, dtype=np.float32)
    r, d = g.shape
    cnt = python_prefix_len(r, int(r - corrupted_count) - 1)
    out = np.empty(d, dtype=np.float32)
    for s in range(0, d, block):
        blk = g[:, s:s + block]
        med = np.median(blk, axis=0).astype(np.float32)
        dev = blk - med[None, :]
        order = np.argsort(np.abs(dev), axis=0, kind='stable')[:cnt]
        kept = np.take_along_axis(dev, order, axis=0).astype(np.float64)
        if cnt:
            mean = (kept.sum(axis=0) / cnt).astype(np.float32)
        else:
            mean = np.full(blk.shape[1], np.nan, dtype=np.float32)
        out[s:s + block] = mean + med
    return out


def bulyan(users_grads, users_count, corrupted_count, dist=None, return_selection=False):
    assert users_count >= 4 * corrupted_count + 3
    g = np.asarray(users_grads)
    if dist is None:
        dist = distance_matrix(g)
    picked = bulyan_selection(dist, users_count, corrupted_count)
    agg = trimmed_mean(g[picked], 2 * corrupted_count)
    if return_selection:
        return agg, picked
    return agg


def drift_vector(rows, num_std):
    """mean - z * population-std per column, fp64 accumulate, fp32 result."""
    r = np.asarray(rows, dtype=np.float64)
    mean = r.mean(axis=0)
    std = r.std(axis=0)
    return (mean - num_std * std).astype(np.float32), mean.astype(np.float32), std.astype(np.float32)
