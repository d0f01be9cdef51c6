#!/usr/bin/env python3
"""Mint golden vectors by running the UNMODIFIED reference in this container.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Reads  /root/reference/{defences,malicious,backdoor,server}.py (imported, never copied; the modules those
two import that are absent from this image -- tensorflow, torchvision via data_sets, user -- are stubbed,
none of their code is on the lines exercised here).
Writes tests/golden/reference_vectors.npz: for each case the seeded input matrix
and whatever the reference returned.  The reference ships no fixtures of its own
(SURVEY.md section 4), so "the reference's output in this image" (numpy 2.2.6,
OpenBLAS 0.3.29) is the pin for both the CPU oracle and the HIP path.
"""
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
import defences as ref_defences  # noqa: E402
import malicious as ref_malicious  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference_neighbours():
    """backdoor.py and server.py import packages this image lacks; stub exactly those for the duration of the import."""
    import types
    stubs = {'data_sets': {'MNIST': 'MNIST', 'CIFAR10': 'CIFAR10'},
             'user': {'flatten_params': None, 'row_into_parameters': None, 'cycle': None},
             'tensorflow': {}}
    for name, attrs in stubs.items():
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)
        sys.modules[name] = mod
    try:
        import backdoor as ref_backdoor
        import server as ref_server
    finally:
        for name in stubs:
            del sys.modules[name]
    return ref_backdoor, ref_server


def import_reference_clients():
    """user.py and data_sets.py with torchvision (absent here) stubbed: the networks and `User.step` need none of it."""
    import types
    tv = types.ModuleType('torchvision')
    tv.datasets = types.ModuleType('torchvision.datasets')
    tv.transforms = types.ModuleType('torchvision.transforms')
    stubs = {'torchvision': tv, 'torchvision.datasets': tv.datasets, 'torchvision.transforms': tv.transforms}
    sys.modules.update(stubs)
    try:
        import data_sets as ref_data_sets
        import user as ref_user
    finally:
        for name in stubs:
            del sys.modules[name]
    return ref_data_sets, ref_user


CLIENT_SAMPLE_ROWS = (0, 57)   # hidden units whose fc1.weight gradient rows are stored in full


def gaussian(seed, n, d):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


def scaled(seed, n, d):
    """Well-separated Krum scores: row i scaled by 1 + 0.5*perm(i)/n."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    s = (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)
    return g * s[:, None]


class FakeUser:
    def __init__(self, grads):
        self.grads = grads
        self.original_params = None
        self.learning_rate = None


def attacked(seed, n, d, m, z=1.5, family=gaussian):
    """Rows 0..m-1 replaced by the reference's own drift vector (malicious.py)."""
    g = family(seed, n, d)
    users = [FakeUser(g[i].copy()) for i in range(m)]
    ref_malicious.DriftAttack(z).attack(users)
    for i in range(m):
        g[i] = users[i].grads
    return g


def bulyan_selection(g, n, f):
    """Replay defences.py:59-68 with the reference's own functions to expose the picks."""
    dist = ref_defences._krum_create_distances(g)
    picks = []
    while len(picks) < n - 2 * f:
        idx = ref_defences.krum(g, n - len(picks), f, dist, True)
        picks.append(idx)
        dist.pop(idx)
        for r in dist:
            dist[r].pop(idx)
    return np.asarray(picks, dtype=np.int64)


def dense(dist_dict, n):
    out = np.full((n, n), np.inf, dtype=np.float32)
    for i, row in dist_dict.items():
        for j, v in row.items():
            out[i, j] = v
    return out


def main():
    out = {}

    def put(case, **kv):
        for k, v in kv.items():
            out['%s/%s' % (case, k)] = np.asarray(v)

    # --- no_defense -------------------------------------------------------------------
    g = gaussian(11, 7, 130)
    put('nodef_7x130', G=g, out=ref_defences.no_defense(g, 7, 1))

    # --- krum ------------------------------------------------------------------------
    for name, g, f in [
        ('krum_iid_10x257', gaussian(21, 10, 257), 2),
        ('krum_scaled_33x1000', scaled(22, 33, 1000), 8),
        ('krum_attacked_12x300', attacked(23, 12, 300, 3), 3),       # identical rows -> exact ties
        ('krum_allsame_6x64', np.tile(gaussian(24, 1, 64), (6, 1)), 1),  # every score ties -> row 1
        ('krum_f0_5x40', gaussian(25, 5, 40), 0),                      # prefix longer than the list
    ]:
        n = len(g)
        put(name, G=g, f=f,
            dist=dense(ref_defences._krum_create_distances(g), n),
            index=ref_defences.krum(g, n, f, return_index=True),
            out=ref_defences.krum(g, n, f))
    g = np.full((4, 8), np.nan, dtype=np.float32)
    put('krum_allnan_4x8', G=g, f=1, index=ref_defences.krum(g, 4, 1, return_index=True))

    # --- trimmed_mean ----------------------------------------------------------------
    for name, g, c in [
        ('tm_odd_11x97', gaussian(31, 11, 97), 2),
        ('tm_even_10x97', gaussian(32, 10, 97), 2),
        ('tm_100x64', gaussian(33, 100, 64), 20),
        ('tm_attacked_20x50', attacked(34, 20, 50, 4), 4),
        ('tm_c0_9x33', gaussian(35, 9, 33), 0),
    ]:
        put(name, G=g, c=c, out=ref_defences.trimmed_mean(g, len(g), c))
    # +a / -a ties at the window edge: the lower row index must win (stable sort)
    col = np.array([0.0, 3.0, -3.0, 1.0, -1.0, 5.0, -5.0], dtype=np.float32)  # median 0
    g = np.stack([col, col[::-1].copy(), np.roll(col, 3)], axis=1)
    for c in (1, 2, 3, 4):
        put('tm_edge_ties_c%d' % c, G=g, c=c, out=ref_defences.trimmed_mean(g, 7, c))
    g = gaussian(36, 6, 20)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        put('tm_kzero_6x20', G=g, c=5, out=ref_defences.trimmed_mean(g, 6, 5))      # k = 0 -> NaN
    put('tm_kneg_6x20', G=g, c=7, out=ref_defences.trimmed_mean(g, 6, 7))           # k = -2 -> [:-2]

    # --- bulyan ----------------------------------------------------------------------
    for name, g, f in [
        ('bulyan_iid_11x200', gaussian(41, 11, 200), 2),
        ('bulyan_boundary_15x120', gaussian(42, 15, 120), 3),            # n == 4f + 3
        ('bulyan_scaled_40x500', scaled(43, 40, 500), 9),
        ('bulyan_attacked_23x150', attacked(44, 23, 150, 5), 5),
        ('bulyan_f0_6x30', gaussian(45, 6, 30), 0),
    ]:
        n = len(g)
        put(name, G=g, f=f, selection=bulyan_selection(g, n, f), out=ref_defences.bulyan(g, n, f))

    # --- attack ----------------------------------------------------------------------
    for name, g, z in [('attack_5x300_z1.5', gaussian(51, 5, 300), 1.5),
                       ('attack_24x100_z0.5', gaussian(52, 24, 100) * 3 + 7, 0.5),
                       ('attack_3x64_z0', gaussian(53, 3, 64), 0.0)]:
        users = [FakeUser(r.copy()) for r in g]
        att = ref_malicious.DriftAttack(z)
        att.attack(users)
        put(name, G=g, z=z, stored_mean=att.grads_mean, stored_stdev=att.grads_stdev,
            user0=users[0].grads, aliased=int(all(u.grads is users[0].grads for u in users)))

    # --- neighbours of the path: backdoor hook, gradient assembly --------------------
    ref_backdoor, ref_server = import_reference_neighbours()
    for name, d, z, lr, spread in [('backdoor_300_z1.5', 300, 1.5, 0.1, 0.3),
                                   ('backdoor_1000_z0.5_faded_lr', 1000, 0.5, 0.1 * 10 / (3 + 10), 2.0),
                                   ('backdoor_64_z0', 64, 0.0, 0.05, 1.0)]:
        rng = np.random.default_rng(60 + d)
        mean = rng.standard_normal(d).astype(np.float32)
        stdev = np.abs(rng.standard_normal(d)).astype(np.float32)
        params = rng.standard_normal(d).astype(np.float32)
        mal = (params + spread * rng.standard_normal(d)).astype(np.float32)
        att = object.__new__(ref_backdoor.BackdoorAttack)   # __init__ builds data loaders; the hook needs none
        att.num_std = z
        seen = {}

        def train(start, _mal=mal, _seen=seen):   # stands in for backdoor.py:56's training loop
            _seen['start'] = start.copy()
            return _mal
        att.train_malicious_network = train
        out_vec = att._attack_grads(mean.copy(), stdev.copy(), params.copy(), lr)
        put(name, mean=mean, stdev=stdev, params=params, mal=mal, z=z, lr=lr, start=seen['start'], out=out_vec)

    class FakeServer:
        pass
    srv = FakeServer()
    rng = np.random.default_rng(71)
    shapes = [(20, 7), (20,), (3, 5, 2), (1,), (13,)]
    per_user = [[rng.standard_normal(sh).astype(np.float32) for sh in shapes] for _ in range(4)]
    srv.users = [FakeUser(np.concatenate([t.flatten() for t in tensors])) for tensors in per_user]   # user.py:92
    srv.users_grads = np.empty((4, srv.users[0].grads.size), dtype=np.float32)                       # server.py:35
    ref_server.Server.collect_gradients(srv)
    put('assemble_4x%d' % srv.users_grads.shape[1], G=srv.users_grads,
        **{'u%d_t%d' % (u, t): per_user[u][t] for u in range(4) for t in range(len(shapes))})

    # --- the client step (user.py:76-92), 3 clients of MnistNet from the same weights ------------------
    import types
    import torch
    ref_data_sets, ref_user = import_reference_clients()
    rng = np.random.default_rng(81)
    n_clients, batch = 3, 5
    weights = (rng.integers(-128, 128, size=79510) / 1024.0).astype(np.float32)      # compressible, exact in fp32
    data = (rng.integers(-64, 64, size=(n_clients, batch, 1, 28, 28)) / 32.0).astype(np.float32)
    target = rng.integers(0, 10, size=(n_clients, batch)).astype(np.int64)
    rows = []
    for c in range(n_clients):
        usr = types.SimpleNamespace(user_id=c, is_malicious=False, momentum=0.9, data_set=ref_data_sets.MNIST,
                                    net=ref_data_sets.MnistNet(), criterion=torch.nn.NLLLoss(),
                                    train_iterator=iter([(torch.from_numpy(data[c]), torch.from_numpy(target[c]))]))
        usr.train = types.MethodType(ref_user.User.train, usr)
        ref_user.User.step(usr, weights, 0.1)
        rows.append(usr.grads)
    rows = np.stack(rows)
    w1 = rows[:, :78400].reshape(n_clients, 100, 784)
    put('clients_mnist_3x5', weights=weights, data=data, target=target,
        fc1_weight_rows=w1[:, CLIENT_SAMPLE_ROWS, :], tail=rows[:, 78400:],          # fc1.bias, fc2.weight, fc2.bias in full
        row_sums=rows.astype(np.float64).sum(axis=1), row_norms=np.sqrt((rows.astype(np.float64) ** 2).sum(axis=1)))

    path = os.path.join(HERE, 'reference_vectors.npz')
    np.savez_compressed(path, **out)
    print('wrote %s: %d arrays, %.1f KiB' % (path, len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
