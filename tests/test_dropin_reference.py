"""The reference's own `main.main` and `Server.defend`, unmodified, driven through this repo's drop-in modules.

`/root/reference` exists only in the build container, which has no GPU; the GPU box has the GPU and no reference.
So the claim "drops into main.py unchanged" (BASELINE.json north_star) is split where the two boxes split it:

  here (marker `reference`)  the unmodified main.py / server.py / user.py run two rounds with `dropin/` first on
        sys.path.  Their `import defences`, `import malicious` bind this repo's modules; every call goes through the
        drop-in signatures, return types and the attacker's rebinding protocol.  Behind the modules sits the real engine
        when an MI355X is visible and otherwise a test double with the engine's method surface computing with the CPU
        oracle -- so what is under test on a CPU box is the boundary, bit for bit, not the kernels.
        The same rounds with the reference's OWN defences/malicious give the trajectory to compare with, and a small
        golden record of it is committed (tests/golden/dropin_rounds.npz, written by this file with
        BYZ_MINT_GOLDEN=1).
  GPU box (marker `gpu`)  tests/test_gpu_parity.py::test_golden_dropin_rounds replays the recorded rounds through the same
        drop-in modules on the real engine and must land on the recorded weights.
"""
import os
import tempfile

import numpy as np
import pytest

from oracle import faithful

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'dropin_rounds.npz')
DEFENCES = ['NoDefense', 'Krum', 'TrimmedMean', 'Bulyan']

pytestmark = pytest.mark.reference


class OracleBackedEngine:
    """Test double with the method surface `defences.py` / `malicious.py` use of `engine.Engine`, computing with the CPU
    oracle.  Lets the drop-in boundary be exercised by the unmodified reference on a box without a GPU."""

    def no_defense(self, g, users_count=None, corrupted_count=None):
        return faithful.no_defense(np.asarray(g), users_count, corrupted_count)

    def pairwise_distances(self, g):
        return faithful.distance_matrix(np.asarray(g))

    def krum(self, g, users_count, corrupted_count, distances=None, return_index=False):
        out = faithful.krum(np.asarray(g), users_count, corrupted_count, distances=distances, return_index=return_index)
        return out        # a view of the caller's matrix, like the reference (and, since round 6, like the engine's host path)

    def trimmed_mean(self, g, users_count=None, corrupted_count=0):
        return faithful.trimmed_mean(np.asarray(g), users_count, corrupted_count)

    def bulyan(self, g, users_count, corrupted_count):
        return faithful.bulyan(np.asarray(g), users_count, corrupted_count)

    def drift_attack(self, rows, num_std, write_back=False):
        rows = np.asarray(rows, dtype=np.float32)
        mean, std = faithful.attack_statistics(rows)
        return mean - num_std * std, mean, std

    def drift_axpy_host(self, mean, std, num_std):
        out = np.array(mean, dtype=np.float32, copy=True)
        out[:] -= num_std * np.asarray(std)[:]
        return out


@pytest.fixture(scope='module')
def engine_in_use():
    """The real engine when this box has an MI355X (then this IS the end-to-end drop-in run), else the double."""
    from attacking_federate_learning_amd import engine as engine_module
    saved = engine_module._default
    kind = 'hip'
    try:
        engine_module._default = None
        engine_module.get_engine()
    except Exception:
        engine_module._default = OracleBackedEngine()
        kind = 'oracle-double'
    yield kind
    engine_module._default = saved


@pytest.fixture(scope='module')
def harness():
    if not os.path.isfile('/root/reference/main.py'):
        pytest.skip('reference checkout not present on this box')
    from tests import reference_harness
    return reference_harness


@pytest.mark.parametrize('defense', DEFENCES)
def test_reference_main_runs_unchanged_on_the_dropin_modules(harness, engine_in_use, defense):
    with tempfile.TemporaryDirectory() as tmp:
        own = harness.run_reference_main(False, defense, os.path.join(tmp, 'own'))
        ours = harness.run_reference_main(True, defense, os.path.join(tmp, 'dropin'))
    # the unmodified driver files in both runs; only `defences` / `malicious` differ
    assert own['modules']['main'] == ours['modules']['main'] == '/root/reference/main.py'
    assert own['modules']['server'] == ours['modules']['server'] == '/root/reference/server.py'
    assert own['modules']['defences'] == '/root/reference/defences.py'
    assert ours['modules']['defences'].endswith('attacking_federate_learning_amd/dropin/defences.py')
    assert ours['modules']['malicious'].endswith('attacking_federate_learning_amd/dropin/malicious.py')
    assert len(own['weights']) == len(ours['weights']) == 2
    for epoch in range(2):
        # same client gradients and the same attack went in ...
        if engine_in_use == 'oracle-double':
            assert np.array_equal(own['users_grads'][epoch], ours['users_grads'][epoch])
            assert np.array_equal(own['weights'][epoch], ours['weights'][epoch]), (defense, epoch)
        else:
            assert np.allclose(own['users_grads'][epoch], ours['users_grads'][epoch], rtol=1e-5, atol=1e-6)
            assert np.allclose(own['weights'][epoch], ours['weights'][epoch], rtol=1e-5, atol=1e-6), (defense, epoch)
    # the attack really happened: the malicious rows are one vector (malicious.py:26-27), the honest ones are not
    g = ours['users_grads'][1]
    assert np.array_equal(g[0], g[1]) and not np.array_equal(g[1], g[2])
    # and the rounds moved the weights
    assert not np.array_equal(own['weights'][0], own['weights'][1])


def test_server_defend_of_the_reference_calls_the_dropin_signature(harness, engine_in_use):
    """`Server.defend` alone (server.py:86-90) on a hand-made server object: the exact call of server.py:87."""
    with harness.reference_imports(True):
        import server as ref_server
        import defences as bound
        srv = ref_server.Server.__new__(ref_server.Server)      # skip __init__ (data loaders): defend needs five fields
        rng = np.random.default_rng(3)
        n, d = 15, 400
        srv.users = [None] * n
        srv.mal_prop, srv.momentum, srv.learning_rate = 0.2, 0.9, 0.1
        srv.users_grads = rng.standard_normal((n, d)).astype(np.float32)
        for name in DEFENCES:
            srv.current_weights = rng.standard_normal(d).astype(np.float32)
            srv.velocity = rng.standard_normal(d).astype(np.float32)
            w0, v0, g0 = srv.current_weights.copy(), srv.velocity.copy(), srv.users_grads.copy()
            srv.defend(name, 0)
            agg = {'NoDefense': faithful.no_defense, 'Krum': faithful.krum, 'TrimmedMean': faithful.trimmed_mean,
                   'Bulyan': faithful.bulyan}[name](g0, n, int(n * 0.2))
            v1 = np.float32(0.9) * v0 - 0.1 * agg
            tol = dict(rtol=0, atol=0) if engine_in_use == 'oracle-double' else dict(rtol=1e-5, atol=1e-6)
            assert np.allclose(srv.velocity, v1, **tol) and np.allclose(srv.current_weights, w0 + v1, **tol)
            assert np.array_equal(srv.users_grads, g0)                      # the callee must not touch the matrix
        assert bound.defend is bound.defend and set(bound.defend) == set(DEFENCES)


def test_mint_or_check_the_golden_rounds(harness):
    """A compact record of the reference's rounds for the GPU box (which has no reference): the matrices the unmodified
    main loop handed to `Server.defend` (client gradients of the reference's own `User.step`, drifted by its own
    `DriftAttack`), cut to 3000 columns, and what the reference's own `Server.defend` (server.py:86-90, its own
    defences.py) makes of them over two rounds with momentum.  tests/test_gpu_parity.py replays exactly that through
    the drop-in modules on the MI355X."""
    with tempfile.TemporaryDirectory() as tmp:
        runs = {d: harness.run_reference_main(False, d, os.path.join(tmp, d)) for d in DEFENCES}
    record = {}
    first = runs['NoDefense']['users_grads'][0]
    for d in DEFENCES:     # round 0 aggregates the same matrix whatever the defence (same seeds, same initial weights)
        assert np.array_equal(runs[d]['users_grads'][0], first)
    n = first.shape[0]
    cols = np.sort(np.random.default_rng(0).choice(first.shape[1], 3000, replace=False))
    record['round0_grads'] = np.ascontiguousarray(first[:, cols])
    rng = np.random.default_rng(1)
    record['weights_start'] = rng.standard_normal(len(cols)).astype(np.float32)
    record['mal_prop'] = np.float64(0.24)
    with harness.reference_imports(False):
        import server as ref_server
        for d in DEFENCES:
            record['%s/round1_grads' % d] = np.ascontiguousarray(runs[d]['users_grads'][1][:, cols])
            srv = ref_server.Server.__new__(ref_server.Server)
            srv.users, srv.mal_prop, srv.momentum, srv.learning_rate = [None] * n, 0.24, 0.9, 0.1
            srv.current_weights = record['weights_start'].copy()
            srv.velocity = np.zeros_like(srv.current_weights)
            for e, key in enumerate(('round0_grads', '%s/round1_grads' % d)):
                srv.users_grads = record[key].copy()
                srv.defend(d, e)
                record['%s/weights%d' % (d, e)] = srv.current_weights.copy()
    if os.environ.get('BYZ_MINT_GOLDEN') == '1' or not os.path.exists(GOLDEN):
        np.savez_compressed(GOLDEN, **record)
    z = np.load(GOLDEN)
    assert sorted(z.files) == sorted(record)
    for key in record:
        assert np.array_equal(z[key], record[key]), key
