"""fp32 restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function cites the reference lines it follows.  The restatement keeps a dense
(N, N) distance matrix instead of the reference's dict-of-dicts, but performs the
same floating-point operations in the same order, so its outputs are bit-identical
to the reference's (checked in tests/test_oracle_vs_reference.py and through the
golden vectors in tests/golden/).

Numerics pinned here:
  * distances are *unsquared* L2 norms of the fp32 difference, via np.linalg.norm
    (sqrt of OpenBLAS sdot)                                  -- defences.py:16-21
  * a Krum score is the *sequential* fp32 sum of the ascending-sorted distances,
    first ``users_count - corrupted_count`` of them (Python ``sum`` over
    np.float32 scalars under numpy>=2)                       -- defences.py:33-34
  * candidates are visited in dict-insertion order 1, 0, 2, 3, ... with a strict
    ``<`` against a running minimum that starts at 1e20 / index -1
                                                             -- defences.py:27-37
  * trimmed_mean keeps the k = rows - corrupted - 1 values closest to the fp32
    median, ties in |x - med| resolved by row order (stable sort), and returns
    np.mean(kept deviations) + med                           -- defences.py:44-52
  * Bulyan feeds the selected rows *in selection order* to trimmed_mean
                                                             -- defences.py:55-70
"""
import numpy as np

SELF = np.float32(np.inf)  # marker on the diagonal: the reference stores no self-distance


def visit_order(n_rows):
    """Key order of the reference's distance dict: 1, 0, 2, 3, ... (defences.py:17-20).

    ``distances[i][j] = distances[j][i] = ...`` touches key i before key j, and the
    first pair is (1, 0).  With fewer than two rows the dict stays empty.
    """
    if n_rows < 2:
        return []
    return [1, 0] + list(range(2, n_rows))


def python_prefix_len(length, stop):
    """Number of elements ``some_list[:stop]`` keeps for a list of ``length`` items."""
    if stop >= 0:
        return min(stop, length)
    return max(length + stop, 0)


def no_defense(users_grads, users_count, corrupted_count):
    """defences.py:13-14 -- column mean in fp32."""
    return np.mean(users_grads, axis=0)


def distance_matrix(users_grads):
    """defences.py:16-21 as a dense symmetric fp32 matrix; diagonal = +inf marker."""
    g = np.asarray(users_grads)
    n = len(g)
    dist = np.full((n, n), SELF, dtype=np.float32)
    for i in range(n):
        for j in range(i):
            d = np.linalg.norm(g[i] - g[j])
            dist[i, j] = d
            dist[j, i] = d
    return dist


def sequential_sum_f32(values):
    """Python ``sum`` over np.float32 scalars: left-to-right fp32 adds (0 + x is exact)."""
    values = np.asarray(values, dtype=np.float32)
    if values.size == 0:
        return 0  # Python's integer zero, as sum([]) returns
    return np.cumsum(values, dtype=np.float32)[-1]


def krum_scores(dist, alive, users_count, corrupted_count):
    """Scores of every live row, defences.py:26,33-34.

    ``alive`` lists the rows still present in the dict, in dict order.  Returns a
    dict row -> score (np.float32, or int 0 for an empty prefix).
    """
    keep = users_count - corrupted_count
    alive_idx = np.asarray(alive, dtype=np.int64)
    scores = {}
    for u in alive:
        others = alive_idx[alive_idx != u]
        errors = np.sort(dist[u, others])  # same multiset order as sorted() for non-NaN input
        cnt = python_prefix_len(len(errors), keep)
        scores[u] = sequential_sum_f32(errors[:cnt])
    return scores


def krum_pick(dist, alive, users_count, corrupted_count):
    """The selection loop of defences.py:27-37 on a dense matrix; returns the index."""
    scores = krum_scores(dist, alive, users_count, corrupted_count)
    best, best_idx = 1e20, -1
    for u in alive:
        if scores[u] < best:
            best, best_idx = scores[u], u
    return best_idx


def krum(users_grads, users_count, corrupted_count, distances=None, return_index=False):
    """defences.py:23-42.  ``distances`` is a dense matrix from distance_matrix()."""
    if not return_index:
        assert users_count >= 2 * corrupted_count + 1, (
            'users_count>=2*corrupted_count + 3', users_count, corrupted_count)
    if distances is None:
        distances = distance_matrix(users_grads)
    idx = krum_pick(distances, visit_order(len(distances)), users_count, corrupted_count)
    if return_index:
        return idx
    return users_grads[idx]


def trimmed_mean_column(column, keep):
    """One iteration of defences.py:48-51; ``keep`` is the raw slice stop (may be <= 0)."""
    med = np.median(column)
    dev = column - med
    order = np.argsort(np.abs(dev), kind='stable')  # == sorted(..., key=abs): stable
    cnt = python_prefix_len(len(dev), keep)
    good = dev[order[:cnt]]
    with np.errstate(all='ignore'):
        return np.mean(good) + med if cnt else np.float32(np.nan) + med


def trimmed_mean(users_grads, users_count, corrupted_count):
    """defences.py:44-52.  Uses the row count of the matrix, not ``users_count``."""
    g = np.asarray(users_grads)
    keep = int(g.shape[0] - corrupted_count) - 1
    out = np.empty((g.shape[1],), g.dtype)
    for i, column in enumerate(g.T):
        out[i] = trimmed_mean_column(column, keep)
    return out


def bulyan_selection(dist, users_count, corrupted_count):
    """The while loop of defences.py:59-68: indices in selection order."""
    set_size = users_count - 2 * corrupted_count
    alive = visit_order(len(dist))
    picked = []
    while len(picked) < set_size:
        idx = krum_pick(dist, alive, users_count - len(picked), corrupted_count)
        if idx == -1:
            # reference: users_grads[-1] is appended, then distances.pop(-1) raises KeyError
            raise KeyError(-1)
        picked.append(idx)
        alive = [u for u in alive if u != idx]
    return picked


def bulyan(users_grads, users_count, corrupted_count, return_selection=False):
    """defences.py:55-70."""
    assert users_count >= 4 * corrupted_count + 3
    g = np.asarray(users_grads)
    picked = bulyan_selection(distance_matrix(g), users_count, corrupted_count)
    agg = trimmed_mean(g[picked], len(picked), 2 * corrupted_count)
    if return_selection:
        return agg, picked
    return agg


def attack_statistics(rows):
    """malicious.py:18-19 -- fp32 column mean and population standard deviation."""
    mean = np.mean(rows, axis=0)
    stdev = np.var(rows, axis=0) ** 0.5
    return mean, stdev


def attack_statistics_sequential(rows):
    """The same two lines as fp32 operations in the order numpy performs them (a reduction over the OUTER axis of a C-ordered
    array is a plain loop over the rows, not a pairwise sum; `** 0.5` on an fp32 array is np.sqrt): the arithmetic
    csrc/column_stats.hip implements.  tests/test_pipeline_golden.py holds it to `attack_statistics` bit for bit."""
    a = np.asarray(rows, dtype=np.float32)
    m = np.float32(a.shape[0])
    s = np.zeros(a.shape[1], dtype=np.float32)        # add.reduce starts from its identity: a column of -0.0 sums to +0.0
    for r in range(a.shape[0]):
        s = s + a[r]
    mean = s / m
    s2 = np.zeros(a.shape[1], dtype=np.float32)
    for r in range(a.shape[0]):
        d = a[r] - mean
        s2 = s2 + d * d
    return mean, np.sqrt(s2 / m)


def drift_vector(rows, num_std):
    """malicious.py:18-24,34-36: the vector every malicious client submits."""
    mean, stdev = attack_statistics(rows)
    if num_std == 0:
        return None
    mean[:] -= num_std * stdev[:]
    return mean


# ---- the steps either side of the path ------------------------------------

def backdoor_initial_params(original_params, learning_rate, grads_mean):
    """backdoor.py:54 -- where the malicious network starts: the parameters after the honest mean step.
    `learning_rate` is a Python float, so numpy keeps the arithmetic in fp32 (weak scalar)."""
    return original_params - learning_rate * grads_mean


def backdoor_attack_grads(grads_mean, grads_stdev, original_params, learning_rate, num_std, mal_net_params):
    """backdoor.py:52-65 with the training loop's result (`mal_net_params`, backdoor.py:56) handed in:
    the gradient that moves the parameters to the malicious network's, clipped to mean +- num_std * std."""
    start = backdoor_initial_params(original_params, learning_rate, grads_mean)
    wanted_params = mal_net_params + learning_rate * grads_mean          # backdoor.py:59
    wanted_grads = (start - wanted_params) / learning_rate               # backdoor.py:60
    band = num_std * grads_stdev
    return np.clip(wanted_grads, grads_mean - band, grads_mean + band)   # backdoor.py:62-63


def assemble_row(users_grads, idx, tensors):
    """user.py:92 then server.py:83 -- a client's per-parameter gradients, flattened and concatenated in
    parameter order, become row `idx` of the server's matrix."""
    users_grads[idx, :] = np.concatenate([np.asarray(t).flatten() for t in tensors])
    return users_grads
