"""The reference's own outputs at BASELINE.json's small configurations (tests/golden/baseline_sizes.npz, minted by
tests/golden/make_golden_baseline.py from the imported reference; inputs regenerated from seeds, not stored).

CPU half: the oracle restatement agrees with what the reference returned at these sizes.
GPU half (`-m gpu`): the HIP path through the drop-in modules -- host numpy in, as main.py hands it over -- and through the
device-resident engine returns the reference's index / selection exactly, its distances to 1e-5 relative and its aggregate to
1e-5 (north_star's bars); the trimmed-mean columns also embedded at scattered positions of a full configs[2] matrix
(N = 1000, D = 1e6).
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import baseline_inputs as inputs  # noqa: E402

from oracle import faithful  # noqa: E402

RTOL = ATOL = 1e-5
KRUM_CASES = [c['name'] for c in inputs.CASES if c['kind'].startswith('krum')]
BULYAN_CASES = [c['name'] for c in inputs.CASES if c['kind'] == 'krum+bulyan']
TM_CASES = [c['name'] for c in inputs.CASES if c['kind'] == 'trimmed_mean']
BY_NAME = {c['name']: c for c in inputs.CASES}


@pytest.fixture(scope='module')
def baseline():
    z = np.load(os.path.join(HERE, 'golden', 'baseline_sizes.npz'))
    cases = {}
    for key in z.files:
        case, field = key.split('/', 1)
        cases.setdefault(case, {})[field] = z[key]
    return cases


_made = {}


def seeded(name, baseline):
    """The case's seeded matrix (before the attack), checked against the stored checksum."""
    if name not in _made:
        g = inputs.make(BY_NAME[name])
        assert np.array_equal(inputs.checksum(g), baseline[name]['checksum']), \
            'this box regenerates another input stream than the one the golden outputs were minted on'
        _made.clear()            # one matrix at a time (32 MB each)
        _made[name] = g
    return _made[name].copy()


def close(a, b, rtol=RTOL, atol=ATOL):
    return np.allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol, equal_nan=True)


def off_diagonal_close(dist, want, rtol):
    off = ~np.eye(len(want), dtype=bool)
    zero = want[off] == 0
    return np.all(dist[off][zero] == 0) and np.allclose(dist[off], want[off], rtol=rtol, atol=0.0)


# ---- CPU: the oracle at these sizes ---------------------------------------------------------------------------------
def attacked_by_oracle(name, baseline):
    case, g = BY_NAME[name], seeded(name, baseline)
    if case.get('attack'):
        drift = faithful.drift_vector(g[:case['attack']], case['z'])
        cols = baseline[name]['drift_cols']
        assert np.array_equal(drift[cols], baseline[name]['drift'])      # bit for bit the reference's vector
        g[:case['attack']] = drift
    return g


@pytest.mark.parametrize('name', KRUM_CASES)
def test_oracle_krum_at_baseline_sizes(baseline, name):
    case, want = BY_NAME[name], baseline[name]
    g = attacked_by_oracle(name, baseline)
    dist = faithful.distance_matrix(g)
    # OpenBLAS splits a 79,510-element sdot over its threads: the last bits of a distance depend on the thread count
    assert off_diagonal_close(dist, want['dist'], 1e-5)
    assert faithful.krum(g, case['n'], case['f'], return_index=True) == int(want['index'])
    assert faithful.krum_pick(want['dist'], faithful.visit_order(case['n']), case['n'], case['f']) == int(want['index'])
    if case['kind'] == 'krum+bulyan':
        assert faithful.bulyan_selection(want['dist'], case['n'], case['f']) == want['selection'].tolist()


@pytest.mark.parametrize('name', TM_CASES)
def test_oracle_trimmed_mean_at_baseline_sizes(baseline, name):
    case = BY_NAME[name]
    g = attacked_by_oracle(name, baseline)
    assert np.array_equal(faithful.trimmed_mean(g, case['n'], case['c']), baseline[name]['out'])


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def defences(eng):
    from attacking_federate_learning_amd import defences
    return defences


def attacked_on_the_gpu(name, baseline, eng):
    """The attacked matrix the reference saw, made by OUR attack (malicious.py:10-27 through the C ABI) and by nothing else:
    the vector the defences then get is the GPU's own.  It has to be the reference's to the BIT: with 24 identical values v in a
    column of Bulyan's 52 selected rows the median is (v + h) / 2, h the next value above -- v's copies and h are exactly equally
    far from it in real arithmetic, fp32 rounding decides whether the 3-value window is {v, v, v} or {h, v, v}, and a drift vector
    one ulp off flips that in a few columns by (h - v) / 3.  Rounds 1-4 computed the statistics in fp64 (1e-5 close) and this
    helper substituted the oracle's vector (missing 4); since round 5 the kernel is numpy's arithmetic operation by
    operation and the substitution is gone."""
    case, g = BY_NAME[name], seeded(name, baseline)
    if case.get('attack'):
        m = case['attack']
        drift, _, _ = eng.drift_attack(g[:m], case['z'])
        drift = np.asarray(drift)
        cols = baseline[name]['drift_cols']
        assert np.array_equal(drift[cols], baseline[name]['drift'])                     # the reference's stored columns
        assert np.array_equal(drift, faithful.drift_vector(g[:m], case['z']))           # every column, against the oracle
        g[:m] = drift
    return g


@pytest.mark.gpu
@pytest.mark.parametrize('name', KRUM_CASES)
def test_gpu_krum_at_baseline_sizes(eng, defences, baseline, name):
    case, want = BY_NAME[name], baseline[name]
    n, f = case['n'], case['f']
    g = attacked_on_the_gpu(name, baseline, eng)
    # the selection kernels on the reference's own distance matrix
    assert eng.krum_select(want['dist'], n, f) == int(want['index'])
    # end to end, host numpy in (what server.py:87 passes), the small-N path
    assert defences.krum(g, n, f, return_index=True) == int(want['index'])
    assert np.array_equal(defences.krum(g, n, f), g[int(want['index'])])
    dist = defences._krum_create_distances(g).numpy()
    assert off_diagonal_close(dist, want['dist'], 1e-5)
    assert np.all(np.isinf(np.diag(dist))) and np.array_equal(dist, dist.T)


@pytest.mark.gpu
@pytest.mark.parametrize('name', BULYAN_CASES)
def test_gpu_bulyan_at_baseline_sizes(eng, defences, baseline, name):
    case, want = BY_NAME[name], baseline[name]
    n, f = case['n'], case['f']
    g = attacked_on_the_gpu(name, baseline, eng)
    assert list(eng.bulyan_select(want['dist'], n, f)) == want['selection'].tolist()
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    assert list(sel) == want['selection'].tolist()
    assert close(np.asarray(out)[want['out_cols']], want['out'])
    assert close(defences.bulyan(g, n, f)[want['out_cols']], want['out'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', BULYAN_CASES)
def test_gpu_device_resident_at_baseline_sizes(eng, baseline, name):
    """The same through torch CUDA tensors (nothing crosses PCIe: what bench.py's c2 line times)."""
    torch = pytest.importorskip('torch')
    case, want = BY_NAME[name], baseline[name]
    n, f = case['n'], case['f']
    g = torch.from_numpy(attacked_on_the_gpu(name, baseline, eng)).cuda()
    assert eng.krum(g, n, f, return_index=True) == int(want['index'])
    row = eng.krum(g, n, f)
    assert torch.equal(row, g[int(want['index'])])
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    assert sel.cpu().tolist() == want['selection'].tolist()
    assert close(out.cpu().numpy()[want['out_cols']], want['out'])


@pytest.mark.gpu
@pytest.mark.parametrize('name', TM_CASES)
def test_gpu_trimmed_mean_at_baseline_sizes(eng, defences, baseline, name):
    case, want = BY_NAME[name], baseline[name]
    g = attacked_on_the_gpu(name, baseline, eng)
    assert close(defences.trimmed_mean(g, case['n'], case['c']), want['out'])


@pytest.mark.gpu
def test_gpu_trimmed_mean_columns_inside_the_full_config3_matrix(eng, baseline):
    """configs[2] whole (N = 1000, D = 1e6, 4 GB on the device) with the 512 seeded columns planted at scattered positions:
    the output at those positions is the reference's."""
    torch = pytest.importorskip('torch')
    name = 'c3_tm_1000x512'
    case, want = BY_NAME[name], baseline[name]
    cols = seeded(name, baseline)
    gen = torch.Generator(device='cuda').manual_seed(77)
    g = torch.randn((case['n'], 1_000_000), generator=gen, device='cuda', dtype=torch.float32)
    where = np.sort(np.random.default_rng(78).choice(1_000_000, size=case['d'], replace=False))
    g[:, torch.from_numpy(where).cuda()] = torch.from_numpy(cols).cuda()
    out = eng.trimmed_mean(g, case['n'], case['c'])
    assert close(out.cpu().numpy()[where], want['out'])
