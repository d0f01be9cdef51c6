"""The attack -> defence pipeline against the reference with NOTHING substituted (missing 4 / weak 1).

tests/golden/pipeline_attack.npz holds the unmodified reference's drift vector, standard deviation, Krum index, Bulyan selection
and aggregates IN FULL for seeded inputs (tests/golden/make_golden_pipeline.py).  CPU half: the oracle, and the
operation-by-operation model of numpy's mean / var that the HIP kernel implements, reproduce them bit for bit.  GPU half: the
HIP attack returns the reference's BITS (np.array_equal on the bit patterns, not allclose), and the defences run on the GPU's
OWN vector.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden_pipeline as minted  # noqa: E402

from oracle import faithful  # noqa: E402

BY_NAME = {c['name']: c for c in minted.CASES}
NAMES = list(BY_NAME)


@pytest.fixture(scope='module')
def pipeline():
    z = np.load(os.path.join(HERE, 'golden', 'pipeline_attack.npz'))
    cases = {}
    for key in z.files:
        case, field = key.split('/', 1)
        cases.setdefault(case, {})[field] = z[key]
    return cases


def seeded(name, pipeline):
    g = minted.pipeline_inputs(BY_NAME[name])
    assert np.array_equal(minted.checksum(g), pipeline[name]['checksum']), 'this box regenerates another input stream'
    return g


def same_bits(a, b):
    """Equal as bit patterns, except that any NaN equals any NaN (numpy and the GPU may differ in a NaN's payload)."""
    a, b = np.ascontiguousarray(a, dtype=np.float32), np.ascontiguousarray(b, dtype=np.float32)
    if a.shape != b.shape:
        return False
    both_nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | both_nan))


def close(a, b):
    return np.allclose(np.asarray(a), np.asarray(b), rtol=1e-5, atol=1e-5)


# ---- CPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', NAMES)
def test_oracle_reproduces_the_reference_attack(pipeline, name):
    case, want = BY_NAME[name], pipeline[name]
    g = seeded(name, pipeline)
    rows = g[:case['m']]
    mean, stdev = faithful.attack_statistics(rows)
    assert same_bits(mean, want['mean']) and same_bits(stdev, want['stdev'])
    assert same_bits(faithful.drift_vector(rows, case['z']), want['drift'])


@pytest.mark.parametrize('name', NAMES)
def test_sequential_model_is_numpys_arithmetic(pipeline, name):
    """The kernel's arithmetic, written out: sequential fp32 sum in row order, / float(m), squared fp32 deviations summed in
    row order, / float(m), sqrt.  Equal to np.mean / np.var ** 0.5 to the bit -- on the reference's stored vectors."""
    case, want = BY_NAME[name], pipeline[name]
    rows = seeded(name, pipeline)[:case['m']]
    mean, stdev = faithful.attack_statistics_sequential(rows)
    assert same_bits(mean, want['mean']) and same_bits(stdev, want['stdev'])
    assert same_bits(mean - np.float32(case['z']) * stdev, want['drift'])


@pytest.mark.parametrize('m,d,seed', [(1, 33, 1), (2, 1000, 2), (3, 257, 3), (24, 4096, 4), (100, 700, 5), (777, 129, 6)])
def test_sequential_model_on_more_shapes(m, d, seed):
    rng = np.random.default_rng(seed)
    rows = (rng.standard_normal((m, d)) * rng.uniform(0.01, 30, size=(1, d)) + rng.uniform(-5, 5, size=(1, d))).astype(np.float32)
    rows[:, 0] = 0.0                  # a constant column: std exactly 0
    rows[:, 1] = -0.0
    if d > 8:
        rows[:, 2] = np.float32(1e-30)   # squares underflow to subnormals
        rows[0, 3] = np.inf
        rows[-1, 4] = np.nan
    with np.errstate(all='ignore'):
        want_mean, want_std = faithful.attack_statistics(rows)
        mean, stdev = faithful.attack_statistics_sequential(rows)
    assert same_bits(mean, want_mean) and same_bits(stdev, want_std)
    assert not np.signbit(mean[1]) and not np.signbit(want_mean[1])      # a column of -0.0: the chain starts from +0.0


def test_oracle_pipeline_reproduces_the_reference(pipeline):
    name = 'p_c2_100x21840'
    case, want = BY_NAME[name], pipeline[name]
    g = seeded(name, pipeline)
    g[:case['m']] = faithful.drift_vector(g[:case['m']].copy(), case['z'])
    assert faithful.krum(g, case['n'], case['f'], return_index=True) == int(want['index'])
    out, sel = faithful.bulyan(g, case['n'], case['f'], return_selection=True)
    assert list(sel) == want['selection'].tolist()
    assert close(out, want['bulyan'])        # distances: OpenBLAS's thread count decides their last bits
    name = 'p_tm_1000x384'
    case, want = BY_NAME[name], pipeline[name]
    g = seeded(name, pipeline)
    g[:case['m']] = faithful.drift_vector(g[:case['m']].copy(), case['z'])
    assert same_bits(faithful.trimmed_mean(g, case['n'], case['c']), want['trimmed_mean'])


# ---- GPU ----------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_gpu_attack_is_the_reference_bit_for_bit(eng, pipeline, name):
    case, want = BY_NAME[name], pipeline[name]
    rows = seeded(name, pipeline)[:case['m']]
    drift, mean, std = eng.drift_attack(rows, case['z'])                # host numpy in (malicious.py:14-19)
    assert same_bits(mean, want['mean']) and same_bits(std, want['stdev']) and same_bits(drift, want['drift'])
    torch = pytest.importorskip('torch')
    dev = torch.from_numpy(rows).cuda()
    drift, mean, std = eng.drift_attack(dev, case['z'], write_back=True)   # device-resident, rows overwritten
    assert same_bits(mean.cpu().numpy(), want['mean']) and same_bits(std.cpu().numpy(), want['stdev'])
    assert same_bits(drift.cpu().numpy(), want['drift'])
    assert torch.equal(dev, drift[None, :].expand_as(dev))
    # a strided view (a column slice of a wider matrix, rows not 16-byte aligned): the one-column-per-thread kernel
    wide = torch.zeros((case['m'], rows.shape[1] + 3), device='cuda')
    wide[:, 1:-2] = torch.from_numpy(rows).cuda()
    drift, mean, std = eng.drift_attack(wide[:, 1:-2], case['z'])
    assert same_bits(std.cpu().numpy(), want['stdev']) and same_bits(drift.cpu().numpy(), want['drift'])


@pytest.mark.gpu
def test_gpu_dropin_attack_class_is_the_reference_bit_for_bit(eng, pipeline):
    from attacking_federate_learning_amd import malicious

    class User:
        def __init__(self, grads):
            self.grads, self.original_params, self.learning_rate = grads, None, None

    name = 'p_stats_240x4099'
    case, want = BY_NAME[name], pipeline[name]
    g = seeded(name, pipeline)
    users = [User(g[i].copy()) for i in range(case['m'])]
    att = malicious.DriftAttack(case['z'])
    att.attack(users)
    assert same_bits(users[0].grads, want['drift']) and same_bits(att.grads_stdev, want['stdev'])
    assert same_bits(att.grads_mean, want['drift'])          # malicious.py:35 subtracts in place: the mean IS the drift afterwards
    assert all(u.grads is users[0].grads for u in users)


@pytest.mark.gpu
def test_gpu_pipeline_attack_then_krum_and_bulyan(eng, pipeline):
    """configs[1] under the attack, end to end on the GPU's own drift vector: index and selection are the reference's, the
    aggregate within 1e-5 in every one of the 21,840 columns."""
    from attacking_federate_learning_amd import defences
    name = 'p_c2_100x21840'
    case, want = BY_NAME[name], pipeline[name]
    n, f, m = case['n'], case['f'], case['m']
    g = seeded(name, pipeline)
    drift, _, _ = eng.drift_attack(g[:m], case['z'])
    g[:m] = np.asarray(drift)                              # NOT the oracle's vector
    assert defences.krum(g, n, f, return_index=True) == int(want['index'])
    assert close(defences.krum(g, n, f), want['krum'])
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    assert list(sel) == want['selection'].tolist()
    assert close(out, want['bulyan'])
    assert close(defences.bulyan(g, n, f), want['bulyan'])
    # device-resident, the attack writing its vector into the rows on the device
    torch = pytest.importorskip('torch')
    dev = torch.from_numpy(seeded(name, pipeline)).cuda()
    eng.drift_attack(dev[:m], case['z'], write_back=True)
    assert eng.krum(dev, n, f, return_index=True) == int(want['index'])
    out, sel = eng.bulyan(dev, n, f, return_selection=True)
    assert sel.cpu().tolist() == want['selection'].tolist()
    assert close(out.cpu().numpy(), want['bulyan'])


@pytest.mark.gpu
def test_gpu_pipeline_attack_then_trimmed_mean(eng, pipeline):
    from attacking_federate_learning_amd import defences
    name = 'p_tm_1000x384'
    case, want = BY_NAME[name], pipeline[name]
    g = seeded(name, pipeline)
    drift, _, _ = eng.drift_attack(g[:case['m']], case['z'])
    g[:case['m']] = np.asarray(drift)
    assert close(defences.trimmed_mean(g, case['n'], case['c']), want['trimmed_mean'])


@pytest.mark.gpu
@pytest.mark.parametrize('m,d,z', [(1, 10, 1.5), (24, 79510, 1.5), (240, 5000, 0.7), (5, 1 << 20, 1.5), (2400, 3000, 1.5),
                                   (24, 1 << 21, 1.1), (77, 524288 + 5, 2.0), (2560, 640, 1.5), (2561, 100, 1.5),
                                   (33, 2048, 1.5), (1281, 96, 0.3), (65, 33, 1.5), (641, 4099, 1.5), (640, 2048 + 31, 1.5),
                                   (2400, 8192 * 3 + 17, 1.5), (1999, 12345, 0.9), (320, 100000, 1.5), (81, 70000, 1.5)])
def test_gpu_attack_statistics_bit_for_bit_at_sizes(eng, m, d, z, monkeypatch):
    """Every shape class of csrc/column_stats.hip: the two-pass kernel (m <= 64, m > 2560, fewer than 32 columns), the
    register-resident kernel with four waves (m <= 640) and with sixteen (m <= 2560), every count of row blocks per segment,
    ragged last tiles -- and the two kernels against each other (BYZ_ATTACK_RESIDENT=0)."""
    rng = np.random.default_rng(9000 + m)
    g = (rng.standard_normal((m, d)) * 2 + 0.5).astype(np.float32)
    g[:, 0] = 0.25
    if d > 4:
        g[:, 1] = -0.0
        g[:, 2] = np.float32(3e-23)
    drift, mean, std = eng.drift_attack(g, z)
    want_mean, want_std = faithful.attack_statistics(g)
    assert same_bits(mean, want_mean) and same_bits(std, want_std)
    assert same_bits(drift, faithful.drift_vector(g, z))
    monkeypatch.setenv('BYZ_ATTACK_RESIDENT', '0')
    drift2, mean2, std2 = eng.drift_attack(g, z)
    assert same_bits(mean2, mean) and same_bits(std2, std) and same_bits(drift2, drift)


@pytest.mark.gpu
@pytest.mark.parametrize('m', [66, 130, 200, 260, 330, 400, 450, 520, 580, 640, 700, 1000, 1300, 1600, 1900, 2200, 2500, 2560])
def test_gpu_resident_attack_every_row_block_count(eng, m):
    """RB = 1 .. 10 for both wave counts, device-resident with a leading dimension that is not a multiple of 32 (row segments
    straddle cache lines) and a ragged last tile."""
    torch = pytest.importorskip('torch')
    d = 1000 + m % 37
    rng = np.random.default_rng(9300 + m)
    wide = (rng.standard_normal((m, d + 5)) * 3 - 1).astype(np.float32)
    dev = torch.from_numpy(wide).cuda()
    drift, mean, std = eng.drift_attack(dev[:, 2:d + 2], 1.5)
    eng.check()
    rows = wide[:, 2:d + 2]
    want_mean, want_std = faithful.attack_statistics(rows)
    assert same_bits(mean.cpu().numpy(), want_mean) and same_bits(std.cpu().numpy(), want_std)
    assert same_bits(drift.cpu().numpy(), faithful.drift_vector(rows, 1.5))


@pytest.mark.gpu
@pytest.mark.parametrize('m,d', [(300, 4099), (2400, 3000)])
def test_gpu_resident_attack_redo_path_gives_the_same_bits(eng, m, d, monkeypatch):
    """A wave of the register-resident kernel that never gets its turn sets the call's redo word and the two-pass kernel queued
    behind it recomputes every column (round 5 returned BYZ_OK with an invalid vector and reported it later, from
    another call).  The spin's bound has never been seen to trigger, so the word is forced (BYZ_ATTACK_FORCE_REDO=1): the
    gated launch must then overwrite all three vectors with the same bits, and leave no status behind."""
    rng = np.random.default_rng(9400 + m)
    g = (rng.standard_normal((m, d)) * 2 + 0.5).astype(np.float32)
    drift, mean, std = eng.drift_attack(g, 1.5)
    monkeypatch.setenv('BYZ_ATTACK_FORCE_REDO', '1')
    drift2, mean2, std2 = eng.drift_attack(g, 1.5)
    eng.check()
    assert same_bits(mean2, mean) and same_bits(std2, std) and same_bits(drift2, drift)
    assert same_bits(drift2, faithful.drift_vector(g, 1.5))


@pytest.mark.gpu
def test_gpu_attack_statistics_with_non_finite_rows(eng):
    rng = np.random.default_rng(9200)
    for m in (40, 300, 900):          # the two-pass kernel, the resident one with four waves, with sixteen
        g = rng.standard_normal((m, 300)).astype(np.float32)
        g[3, 5], g[7, 6], g[0, 7], g[m - 1, 8] = np.inf, -np.inf, np.nan, np.nan
        g[:, 9] = np.float32(3e38)        # the sum overflows
        g[:, 10] = -0.0
        with np.errstate(all='ignore'):
            want_mean, want_std = faithful.attack_statistics(g)
        _, mean, std = eng.drift_attack(g, 1.5)
        assert same_bits(mean, want_mean) and same_bits(std, want_std)


@pytest.mark.gpu
@pytest.mark.parametrize('n,d', [(10, 79510), (100, 21840), (1000, 4099), (3, 1 << 20), (1, 7)])
def test_gpu_no_defense_is_numpys_mean_bit_for_bit(eng, n, d):
    """defences.py:13-14: np.mean(axis=0) is the same sequential fp32 chain."""
    rng = np.random.default_rng(9100 + n)
    g = (rng.standard_normal((n, d)) + 0.125).astype(np.float32)
    assert same_bits(eng.no_defense(g, n, 0), faithful.no_defense(g, n, 0))


@pytest.mark.gpu
@pytest.mark.parametrize('m,cuts,d,z', [(24, (7, 15), 21840, 1.5), (2400, (1250, 2400 - 1), 4099, 1.5), (240, (1,), 79510, 0.7),
                                        (100, (33, 34, 99), 1 << 20, 1.5)])
def test_gpu_attack_chain_through_several_owners_is_the_unsharded_attack(eng, m, cuts, d, z):
    """The clients layout's attack (sharded.drift_attack_clients): the malicious rows sit on several ranks, each continues the
    previous one's running sums over its own rows (byz_column_chain_dev), the last ends the chain (byz_column_finish_dev).
    Looped over the owners on one GPU: bit for bit numpy on the stacked rows, and the unsharded kernels' result."""
    torch = pytest.importorskip('torch')
    rng = np.random.default_rng(9400 + m)
    rows = (rng.standard_normal((m, d)) * 1.5 + 0.3).astype(np.float32)
    dev = torch.from_numpy(rows).cuda()
    bounds = [0] + list(cuts) + [m]
    pieces = [dev[a:b] for a, b in zip(bounds[:-1], bounds[1:]) if b > a]

    def walk(mean):
        carry = None
        for piece in pieces:
            carry = eng.column_chain(piece, carry=carry, mean=mean)
        return carry

    mean = eng.column_finish(m, z, sum=walk(None))
    std, drift = eng.column_finish(m, z, sumsq=walk(mean), mean=mean)
    eng.check()
    want_mean, want_std = faithful.attack_statistics(rows)
    assert same_bits(mean.cpu().numpy(), want_mean) and same_bits(std.cpu().numpy(), want_std)
    assert same_bits(drift.cpu().numpy(), faithful.drift_vector(rows, z))
    d2, m2, s2 = eng.drift_attack(dev, z)
    assert torch.equal(d2, drift) and torch.equal(m2, mean) and torch.equal(s2, std)
