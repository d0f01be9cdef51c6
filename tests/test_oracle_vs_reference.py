"""Randomised bit-for-bit checks of oracle.faithful against the reference itself.

Runs only where /root/reference exists (the build container); the GPU box relies on
the golden vectors instead.
"""
import numpy as np
import pytest

from oracle import faithful, ideal

pytestmark = pytest.mark.reference


class FakeUser:
    def __init__(self, grads):
        self.grads, self.original_params, self.learning_rate = grads, None, None


def ref_selection(ref, g, n, f):
    dist = ref._krum_create_distances(g)
    picks = []
    while len(picks) < n - 2 * f:
        idx = ref.krum(g, n - len(picks), f, dist, True)
        picks.append(idx)
        dist.pop(idx)
        for r in dist:
            dist[r].pop(idx)
    return picks


def matrices(seed, n, d, kind):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    if kind == 'dup':           # blocks of identical rows, like the drift attack produces
        g[: n // 3] = g[0]
    elif kind == 'quantised':   # many exact ties in every column
        g = np.round(g * 2).astype(np.float32) / 2
    return g


@pytest.mark.parametrize('kind', ['iid', 'dup', 'quantised'])
@pytest.mark.parametrize('n,d,f', [(2, 5, 0), (3, 17, 1), (9, 64, 2), (16, 33, 3), (31, 257, 7)])
def test_krum_and_distances(reference_modules, kind, n, d, f):
    ref = reference_modules['defences']
    g = matrices(100 + n, n, d, kind)
    dist_ref = ref._krum_create_distances(g)
    assert list(dist_ref.keys()) == faithful.visit_order(n)
    dist = faithful.distance_matrix(g)
    for i in dist_ref:
        for j, v in dist_ref[i].items():
            assert dist[i, j] == v and type(v) is np.float32
    assert faithful.krum(g, n, f, return_index=True) == ref.krum(g, n, f, return_index=True)
    if n >= 2 * f + 1:
        assert np.array_equal(faithful.krum(g, n, f), ref.krum(g, n, f))


@pytest.mark.parametrize('kind', ['iid', 'dup', 'quantised'])
@pytest.mark.parametrize('n,d,c', [(4, 9, 1), (7, 40, 2), (10, 33, 2), (25, 19, 6), (64, 12, 15)])
def test_trimmed_mean(reference_modules, kind, n, d, c):
    ref = reference_modules['defences']
    g = matrices(200 + n, n, d, kind)
    want = ref.trimmed_mean(g, n, c)
    assert np.array_equal(faithful.trimmed_mean(g, n, c), want)
    assert np.allclose(ideal.trimmed_mean(g, c), want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('kind', ['iid', 'dup', 'quantised'])
@pytest.mark.parametrize('n,d,f', [(3, 8, 0), (7, 30, 1), (11, 50, 2), (19, 21, 4), (27, 100, 6)])
def test_bulyan(reference_modules, kind, n, d, f):
    ref = reference_modules['defences']
    g = matrices(300 + n, n, d, kind)
    agg, picked = faithful.bulyan(g, n, f, return_selection=True)
    assert picked == ref_selection(ref, g, n, f)
    assert np.array_equal(agg, ref.bulyan(g, n, f))


def test_assertions_match(reference_modules):
    ref = reference_modules['defences']
    g = matrices(1, 6, 10, 'iid')
    for fn_ref, fn in ((ref.krum, faithful.krum), (ref.bulyan, faithful.bulyan)):
        with pytest.raises(AssertionError):
            fn_ref(g, 6, 3)
        with pytest.raises(AssertionError):
            fn(g, 6, 3)
    # the Krum assert is skipped when only the index is requested (defences.py:24)
    assert faithful.krum(g, 6, 3, return_index=True) == ref.krum(g, 6, 3, return_index=True)


@pytest.mark.parametrize('m,d,z', [(1, 10, 1.5), (4, 77, 1.5), (13, 300, 0.3), (5, 20, 0.0)])
def test_attack(reference_modules, m, d, z):
    ref = reference_modules['malicious']
    g = matrices(400 + m, m, d, 'iid') * 2 + 1
    users = [FakeUser(r.copy()) for r in g]
    att = ref.DriftAttack(z)
    att.attack(users)
    vec = faithful.drift_vector(g, z)
    if z == 0:
        assert vec is None and np.array_equal(users[0].grads, g[0])
    else:
        assert np.array_equal(vec, users[0].grads)
    assert np.array_equal(faithful.attack_statistics(g)[1], att.grads_stdev)


@pytest.fixture(scope='module')
def reference_backdoor(reference_modules):
    """backdoor.py imports `malicious` (the reference's, handed in here), plus `data_sets` and `user`, which
    need torchvision: those two are stubbed for the duration of the import and none of their code runs."""
    import importlib.util
    import os
    import sys
    import types
    path = os.path.join('/root/reference', 'backdoor.py')
    saved = {name: sys.modules.get(name) for name in ('malicious', 'data_sets', 'user')}
    try:
        sys.modules['malicious'] = reference_modules['malicious']
        for name, attrs in (('data_sets', {'MNIST': 'MNIST', 'CIFAR10': 'CIFAR10'}),
                            ('user', {'flatten_params': None, 'row_into_parameters': None, 'cycle': None})):
            mod = types.ModuleType(name)
            mod.__dict__.update(attrs)
            sys.modules[name] = mod
        spec = importlib.util.spec_from_file_location('reference_backdoor', path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for name, old in saved.items():
            if old is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
    return mod


@pytest.mark.parametrize('d,z,lr', [(1, 1.5, 0.1), (257, 1.5, 0.1), (1000, 0.3, 0.1 * 10 / 13), (4096, 3.0, 0.01)])
def test_backdoor_hook(reference_backdoor, d, z, lr):
    rng = np.random.default_rng(900 + d)
    mean = rng.standard_normal(d).astype(np.float32)
    stdev = np.abs(rng.standard_normal(d)).astype(np.float32)
    params = rng.standard_normal(d).astype(np.float32)
    mal = (params + rng.standard_normal(d)).astype(np.float32)
    att = object.__new__(reference_backdoor.BackdoorAttack)   # the constructor builds data loaders
    att.num_std = z
    starts = []
    att.train_malicious_network = lambda start: (starts.append(start.copy()), mal)[1]
    want = att._attack_grads(mean.copy(), stdev.copy(), params.copy(), lr)
    assert np.array_equal(starts[0], faithful.backdoor_initial_params(params, lr, mean))
    assert np.array_equal(want, faithful.backdoor_attack_grads(mean, stdev, params, lr, z, mal), equal_nan=True)


def test_client_step_against_user_step():
    """oracle.clients against the reference's own User.step / MnistNet (torchvision stubbed for the import only)."""
    import importlib.util
    import os
    import sys
    import types
    if not os.path.isfile('/root/reference/user.py'):
        pytest.skip('reference checkout not present on this box')
    torch = pytest.importorskip('torch')
    from oracle import clients
    tv = types.ModuleType('torchvision')
    tv.datasets, tv.transforms = types.ModuleType('torchvision.datasets'), types.ModuleType('torchvision.transforms')
    stubs = {'torchvision': tv, 'torchvision.datasets': tv.datasets, 'torchvision.transforms': tv.transforms}
    saved = {name: sys.modules.get(name) for name in list(stubs) + ['data_sets']}
    try:
        sys.modules.update(stubs)
        mods = {}
        for name in ('data_sets', 'user'):
            spec = importlib.util.spec_from_file_location(name if name == 'data_sets' else 'reference_user',
                                                          os.path.join('/root/reference', name + '.py'))
            mods[name] = importlib.util.module_from_spec(spec)
            if name == 'data_sets':
                sys.modules['data_sets'] = mods[name]      # user.py does `import data_sets`
            spec.loader.exec_module(mods[name])
    finally:
        for name, old in saved.items():
            if old is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
    ref_data_sets, ref_user = mods['data_sets'], mods['user']
    rng = np.random.default_rng(7)
    weights = (0.1 * rng.standard_normal(79510)).astype(np.float32)
    data = rng.standard_normal((4, 9, 1, 28, 28)).astype(np.float32)
    target = rng.integers(0, 10, size=(4, 9))
    want = []
    for c in range(4):
        usr = types.SimpleNamespace(user_id=c, is_malicious=False, momentum=0.9, data_set=ref_data_sets.MNIST,
                                    net=ref_data_sets.MnistNet(), criterion=torch.nn.NLLLoss(),
                                    train_iterator=iter([(torch.from_numpy(data[c]), torch.from_numpy(target[c]))]))
        usr.train = types.MethodType(ref_user.User.train, usr)
        ref_user.User.step(usr, weights, 0.1)
        want.append(usr.grads)
    got = clients.all_client_gradients(clients.MnistNet(), weights, torch.from_numpy(data), torch.from_numpy(target))
    assert np.array_equal(got, np.stack(want))      # same torch ops in the same order: bit-identical
