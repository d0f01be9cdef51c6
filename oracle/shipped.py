"""The reference's loops as shipped: Python dicts, `sorted`, `sum`, a `key=abs` lambda.  TEST INFRASTRUCTURE ONLY.

`oracle.faithful` states the same arithmetic with vectorised numpy (dense matrix, np.sort, np.cumsum), which is several
times faster than what the reference actually executes.  A CPU baseline labelled "reference as shipped" has to pay what
the reference pays: a dict of dicts of np.float32 scalars (defences.py:16-21), `sorted(d.values())` and Python's `sum`
per candidate (defences.py:32-37), and per coordinate a Python `sorted(..., key=lambda x: abs(x))` (defences.py:48-51).
This module restates exactly those loops, line for line with the cited lines, for bench.py's `cpu_baseline` leg on the
GPU box (where /root/reference does not exist).  Pinned against `oracle.faithful` in tests/test_oracle_golden.py.
"""
from collections import defaultdict

import numpy as np


def create_distances(users_grads):
    """defences.py:16-21."""
    distances = defaultdict(dict)
    for i in range(len(users_grads)):
        for j in range(i):
            distances[i][j] = distances[j][i] = np.linalg.norm(users_grads[i] - users_grads[j])
    return distances


def krum(users_grads, users_count, corrupted_count, distances=None, return_index=False):
    """defences.py:23-42."""
    if not return_index:
        assert users_count >= 2 * corrupted_count + 1, ('users_count>=2*corrupted_count + 3', users_count, corrupted_count)
    non_malicious_count = users_count - corrupted_count
    minimal_error = 1e20
    minimal_error_index = -1
    if distances is None:
        distances = create_distances(users_grads)
    for user in distances.keys():
        errors = sorted(distances[user].values())
        current_error = sum(errors[:non_malicious_count])
        if current_error < minimal_error:
            minimal_error = current_error
            minimal_error_index = user
    if return_index:
        return minimal_error_index
    return users_grads[minimal_error_index]


def trimmed_mean(users_grads, users_count, corrupted_count):
    """defences.py:44-52."""
    number_to_consider = int(users_grads.shape[0] - corrupted_count) - 1
    current_grads = np.empty((users_grads.shape[1],), users_grads.dtype)
    for i, param_across_users in enumerate(users_grads.T):
        med = np.median(param_across_users)
        good_vals = sorted(param_across_users - med, key=lambda x: abs(x))[:number_to_consider]
        current_grads[i] = np.mean(good_vals) + med
    return current_grads


def bulyan(users_grads, users_count, corrupted_count):
    """defences.py:55-70."""
    assert users_count >= 4 * corrupted_count + 3
    set_size = users_count - 2 * corrupted_count
    selection_set = []
    distances = create_distances(users_grads)
    while len(selection_set) < set_size:
        currently_selected = krum(users_grads, users_count - len(selection_set), corrupted_count, distances, True)
        selection_set.append(users_grads[currently_selected])
        distances.pop(currently_selected)
        for remaining_user in distances.keys():
            distances[remaining_user].pop(currently_selected)
    return trimmed_mean(np.array(selection_set), len(selection_set), 2 * corrupted_count)


def dict_from_dense(dist):
    """A dense distance matrix as the dict of dicts `create_distances` would have built (same key order)."""
    n = len(dist)
    distances = defaultdict(dict)
    for i in range(n):
        for j in range(i):
            distances[i][j] = distances[j][i] = dist[i, j]
    return distances
