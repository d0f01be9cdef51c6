"""Parity at the sizes BASELINE.json's GPU configurations run (needs an MI355X).

configs[3] is N = 4000 (f = 960, theta = 2080), configs[4] is N = 10000 (f = 2400, theta = 5200): the selection
kernels then sort rows wider than one pass of the workgroup, the Bulyan loop spans many workgroups, and the
second-stage trimmed mean reads 2080 / 5200 rows through the selection.  The oracle at these sizes is
oracle/scale.py (C restatement of defences.py:26-37 and :59-68, pinned against oracle.faithful in
tests/test_oracle_scale.py) -- the reference's own arithmetic (sequential fp32 sums), so given one distance matrix the
indices must be IDENTICAL, not merely within a margin.
"""
import numpy as np
import pytest

from oracle import faithful, ideal, scale

pytestmark = pytest.mark.gpu

MAL_PROP = 0.24      # reference main.py:106
TAU = 16 * np.finfo(np.float32).eps


def close(a, b, rtol=1e-5, atol=1e-5):
    return np.allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol, equal_nan=True)


def point_distances(seed, n, dim=16, identical=0):
    """Distances of n points in `dim` dimensions (fp32, as a client-distance matrix looks to the selection);
    with `identical` > 0 rows 0..identical-1 are one point, like the attack's malicious clients."""
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    pts *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if identical:
        pts[:identical] = pts[:identical].mean(axis=0)
    p64 = pts.astype(np.float64)
    sq = (p64 * p64).sum(1)
    dist = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (p64 @ p64.T), 0.0)).astype(np.float32)
    dist = np.minimum(dist, dist.T)
    if identical:
        dist[:identical, :identical] = 0.0
        dist[:identical, :] = dist[0, :]
        dist[:, :identical] = dist[:, [0]]
    np.fill_diagonal(dist, np.inf)
    return dist


def scaled(seed, n, d):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    return g * (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]


# ---- selection on a given distance matrix: the reference's decisions, bit for bit ----------------------
FULL_ORACLE_MAX = 4000     # the whole selection is recomputed by the C oracle up to here (O(theta N^2))


def sampled_picks(theta, every=40):
    return np.unique(np.concatenate([np.arange(0, theta, every), np.arange(min(12, theta)),
                                     np.arange(max(theta - 12, 0), theta)])).astype(np.int32)


def check_selection(dist, n, f, got):
    """Every pick against the reference's arithmetic up to FULL_ORACLE_MAX rows; beyond that every 40th pick plus both
    ends (each check removes the picks before it and runs the reference's full scoring pass, defences.py:26-37)."""
    theta = n - 2 * f
    assert len(got) == theta and len(set(got)) == theta
    if n <= FULL_ORACLE_MAX:
        want = scale.bulyan_selection(dist, n, f)
        assert got == want, 'first difference at pick %d' % next(i for i, (a, b) in enumerate(zip(got, want)) if a != b)
    else:
        picks = sampled_picks(theta)
        bad, first, expected = scale.verify_picks(dist, n, f, got, picks)
        assert bad == 0, 'pick %d: reference picks row %d, got %d (%d of %d sampled picks differ)' % (
            first, expected, got[first], bad, len(picks))


@pytest.mark.parametrize('n', [200, 333, 640, 1000, 1500, 2049, 4000, 8000, 10000])
def test_bulyan_selection_is_the_reference_selection(eng, n):
    f = int(n * MAL_PROP)
    dist = point_distances(4100 + n, n)
    check_selection(dist, n, f, eng.bulyan_select(dist, n, f).tolist())
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)


@pytest.mark.parametrize('n', [700, 4000, 10000])
def test_bulyan_selection_with_the_attack_ties(eng, n):
    """f identical malicious rows (malicious.py:26-27): their scores tie exactly at every pick and the visit order
    1, 0, 2, ... decides, in the reference and here."""
    f = int(n * MAL_PROP)
    dist = point_distances(4200 + n, n, identical=f)
    got = eng.bulyan_select(dist, n, f).tolist()
    check_selection(dist, n, f, got)
    print('N=%d with %d identical rows: %d rows re-scored in fp32 over %d picks' % (n, f, eng.bulyan_rescored(), n - 2 * f))
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)


@pytest.mark.parametrize('n,dim', [(600, 2000), (2500, 3000)])
def test_bulyan_selection_on_concentrated_scores(eng, n, dim):
    """High-dimensional iid points: all scores lie within a fraction of a percent of each other, so many picks are
    decided inside the rounding noise of the reference's sequential fp32 sums -- the case that needs the reference's
    own arithmetic rather than a more accurate one."""
    f = int(n * MAL_PROP)
    rng = np.random.default_rng(4300 + n)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    p64 = pts.astype(np.float64)
    sq = (p64 * p64).sum(1)
    dist = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (p64 @ p64.T), 0.0)).astype(np.float32)
    dist = np.minimum(dist, dist.T)
    np.fill_diagonal(dist, np.inf)
    want = scale.bulyan_selection(dist, n, f)
    ideal_sel = scale.bulyan_selection(dist, n, f, mode='ideal')
    got = eng.bulyan_select(dist, n, f).tolist()
    assert got == want
    # (informational) this is a case where exact fp64 scoring and the reference disagree
    print('picks where fp64 scoring differs from the reference: %d of %d'
          % (sum(a != b for a, b in zip(ideal_sel, want)), len(want)))


def lattice_distances(seed, n, bits=9):
    """Distances on a coarse binary lattice (integer multiples of 2^-bits, many of them equal): a row's sequential fp32 sum
    meets remainders of exactly half an ulp -- round-to-even ties -- at every other step, and whole groups of equal scores."""
    rng = np.random.default_rng(seed)
    dist = (rng.integers(1 << 6, 1 << 12, (n, n)).astype(np.float32) * np.float32(2.0 ** -bits))
    dist = np.minimum(dist, dist.T)
    np.fill_diagonal(dist, np.inf)
    return dist


@pytest.mark.parametrize('n,family', [(333, 'lattice'), (1500, 'lattice'), (3000, 'lattice'), (1500, 'points'), (1500, 'ties'),
                                      (2049, 'zeros'), (700, 'negative'), (700, 'inf'), (1500, 'nan')])
def test_bulyan_rescore_integer_passes_are_the_literal_chain(eng, monkeypatch, n, family):
    """The default re-score evaluates the reference's sequential fp32 sum in integer passes over a table marked with the
    removals (csrc/select.hip); BYZ_BULYAN_RESCORE=plain is the literal chain of additions over bitmap-tested entries.
    Same selections, pick for pick -- on round-to-even ties at every other addition (lattice), exact-zero distances
    between live rows, a matrix with negative entries (which the passes hand back to the chain), and clients at an infinite /
    NaN distance from everybody (non-finite gradients: they sort last and are never picked) -- and both are the C oracle's
    (the reference's loop)."""
    f = int(n * MAL_PROP)
    if family == 'lattice':
        dist = lattice_distances(4400 + n, n)
    elif family == 'points':
        dist = point_distances(4400 + n, n)
    elif family == 'ties':
        dist = point_distances(4400 + n, n, identical=f)
    elif family == 'zeros':           # groups of coincident honest points: live +0.0 entries inside every prefix
        dist = point_distances(4400 + n, n)
        grp = np.arange(n) // 3
        same = grp[:, None] == grp[None, :]
        dist[same] = 0.0
        np.fill_diagonal(dist, np.inf)
    elif family in ('inf', 'nan'):    # clients whose gradients are not finite: every distance to them is +inf / NaN
        dist = point_distances(4400 + n, n)
        for bad in (17, 300, n - 2):
            dist[bad, :] = np.inf if family == 'inf' else np.nan
            dist[:, bad] = np.inf if family == 'inf' else np.nan
        np.fill_diagonal(dist, np.inf)
    else:
        dist = point_distances(4400 + n, n)
        dist[5, 9] = dist[9, 5] = -0.25
        dist[40, 41] = dist[41, 40] = -0.0
    monkeypatch.setenv('BYZ_BULYAN_RESCORE', 'plain')
    plain = eng.bulyan_select(dist, n, f).tolist()
    rescored = eng.bulyan_rescored()
    monkeypatch.delenv('BYZ_BULYAN_RESCORE')
    monkeypatch.setenv('BYZ_BULYAN_BATCH', '0')     # the sequential loop: the same pairs are re-scored, by the passes
    got = eng.bulyan_select(dist, n, f).tolist()
    assert eng.bulyan_rescored() == rescored
    monkeypatch.delenv('BYZ_BULYAN_BATCH')
    # the default (round 6: batches of picks verified together from 1000 rows): the same selection; picks that are decided again
    # behind a roll-back re-score their contenders again
    batched = eng.bulyan_select(dist, n, f).tolist()
    assert batched == got and eng.bulyan_rescored() >= rescored
    assert got == plain, 'first difference at pick %d' % next(i for i, (a, b) in enumerate(zip(got, plain)) if a != b)
    if family in ('inf', 'nan'):      # such a client is never picked while finite ones remain (the reference's scores are inf)
        assert len(set(got)) == n - 2 * f and not ({17, 300, n - 2} & set(got))
    if family not in ('negative', 'nan'):   # (the oracle restates the reference, whose sorted() is undefined with NaN keys)
        check_selection(dist, n, f, got)
    print('%s N=%d: %d rows re-scored over %d picks' % (family, n, rescored, n - 2 * f))


# ---- the margin protocol, both clauses -----------------------------------------------------
# The engine's distances differ from the reference's sdot by fp32 rounding, so on iid data a pick whose fp64 margin is below
# tau = 16 eps can legitimately go to another row.  Clause 1: the picks BEFORE the first such pick equal the fp64 oracle's.
# Clause 2 (round 6; VERDICT r5 missing 3): every pick -- also the ones after the first contested pick, where the engine's
# selection and the oracle's have parted ways and a prefix comparison says nothing -- must be within tau of the optimum OF ITS
# OWN STATE: the engine's selection is replayed in fp64 with the rows it has picked so far removed
# (oracle/scale.py::replay_selection), and score(engine's pick) <= (1 + tau) * min score at every pick.  The number of
# contested picks is ASSERTED against a recorded ceiling (the measured counts are in profiles/r06_margin_protocol.json;
# the ceilings leave room for a box-to-box flip of a tie, not for a regression) and written out for the record.
MARGIN_CEILINGS = {'c4_n4000_d4096': 110, 'long_k_n3000': 80}      # measured (r06b): 82 and 56


def margin_protocol(name, dist64, n, f, sel, ceiling, want=None, margins=None):
    import json
    import os
    d32 = np.ascontiguousarray(dist64, dtype=np.float32)
    excess, margin, argmin = scale.replay_selection(d32, n, f, sel, mode='ideal')
    contested = int((margin <= TAU).sum())
    parted = int((argmin != np.asarray(sel, dtype=np.int32)).sum())
    record = {'case': name, 'n': n, 'f': f, 'picks': len(sel), 'tau': float(TAU), 'contested_picks': contested,
              'picks_that_are_not_the_fp64_argmin': parted, 'max_excess_over_the_optimum': float(excess.max()),
              'ceiling': ceiling}
    if want is not None:
        first = next((i for i, (a, b) in enumerate(zip(sel, want)) if a != b), len(sel))
        record['first_pick_that_differs_from_the_fp64_selection'] = first
        if margins is not None:
            noisy = np.flatnonzero(margins <= TAU)
            record['first_contested_pick_of_the_fp64_selection'] = int(noisy[0]) if len(noisy) else len(sel)
    try:
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, 'margin_protocol.jsonl'), 'a') as fh:
            fh.write(json.dumps(record) + '\n')
    except OSError:
        pass
    print(record)
    # clause 2: within tau of the optimum at EVERY pick; a pick that is not the fp64 argmin must be a contested one
    assert float(excess.max()) <= TAU, record
    assert np.all(margin[argmin != np.asarray(sel, dtype=np.int32)] <= TAU), record
    assert contested <= ceiling, record
    return record


# ---- end to end at configs[3]'s N ------------------------------------------------------------------------
def test_config4_bulyan_end_to_end_n4000(eng):
    """Bulyan N = 4000, f = 960 on a D = 4096 slice: Gram distances (bf16 x 3 split MFMA, chunk-free schedule),
    row sort with n_pad = 4096, the selection loop, the 2080-row second stage through the selection.
    The distances differ from the reference's sdot by fp32 rounding, so picks inside tau are reported."""
    n, d = 4000, 4096
    f = int(n * MAL_PROP)
    g = scaled(4400, n, d)
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    sel = np.asarray(sel).tolist()
    assert len(sel) == n - 2 * f and len(set(sel)) == len(sel)
    dist64 = ideal.distance_matrix(g)
    want, margins = scale.bulyan_selection(dist64.astype(np.float32), n, f, mode='ideal', with_margins=True)
    noisy = np.flatnonzero(margins <= TAU)
    first_noisy = int(noisy[0]) if len(noisy) else len(sel)
    assert sel[:first_noisy] == want[:first_noisy]
    print('N=4000: %d of %d picks have an fp64 margin below tau; prefix checked: %d' % (len(noisy), len(sel), first_noisy))
    margin_protocol('c4_n4000_d4096', dist64, n, f, sel, MARGIN_CEILINGS['c4_n4000_d4096'], want, margins)
    # the second stage is the reference's trimmed mean of exactly the picked rows, in selection order
    cols = np.random.default_rng(0).choice(d, 96, replace=False)
    assert close(np.asarray(out)[cols], faithful.trimmed_mean(g[sel][:, cols], len(sel), 2 * f))
    assert close(out, ideal.trimmed_mean(g[sel], 2 * f))
    # given the GPU's OWN distances the selection must be the reference's, pick for pick
    dist_gpu = eng.pairwise_distances(g).numpy()
    assert np.allclose(dist_gpu[~np.eye(n, dtype=bool)], dist64[~np.eye(n, dtype=bool)], rtol=1e-6)
    assert sel == scale.bulyan_selection(dist_gpu, n, f)


def test_bulyan_end_to_end_through_the_long_k_gram(eng):
    """The arithmetic BENCH's headline actually runs: N >= 2817 and D > 16384 send the Gram through the operands split
    once into two fp16 planes (gram_planes.hip), then the grid selection loop and the row-split second stage.
    Same protocol as above: distances within 1e-6 of fp64, the picks before the first fp64 margin below tau equal to
    the fp64 oracle's, and -- given the GPU's own distances -- the reference's selection pick for pick."""
    n, d = 3000, 3 * 8192 + 64
    f = int(n * MAL_PROP)
    g = scaled(4700, n, d)
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    sel = np.asarray(sel).tolist()
    assert len(sel) == n - 2 * f and len(set(sel)) == len(sel)
    dist64 = ideal.distance_matrix(g)
    dist_gpu = eng.pairwise_distances(g).numpy()
    off = ~np.eye(n, dtype=bool)
    assert np.allclose(dist_gpu[off], dist64[off], rtol=1e-6)
    want, margins = scale.bulyan_selection(dist64.astype(np.float32), n, f, mode='ideal', with_margins=True)
    noisy = np.flatnonzero(margins <= TAU)
    first_noisy = int(noisy[0]) if len(noisy) else len(sel)
    assert sel[:first_noisy] == want[:first_noisy]
    print('N=3000 (f16x2 Gram): %d of %d picks have an fp64 margin below tau; prefix checked: %d' % (len(noisy), len(sel), first_noisy))
    margin_protocol('long_k_n3000', dist64, n, f, sel, MARGIN_CEILINGS['long_k_n3000'], want, margins)
    assert sel == scale.bulyan_selection(dist_gpu, n, f)
    assert close(out, ideal.trimmed_mean(g[sel], 2 * f))


# ---- second stage: row_index in selection order at theta = 2080 and 5200 --------------------------------------
@pytest.mark.parametrize('n,theta,cols', [(4000, 2080, 70), (10000, 5200, 40), (3000, 1537, 33), (6000, 2561, 20)])
def test_trimmed_mean_through_a_selection(eng, n, theta, cols):
    rng = np.random.default_rng(4500 + n)
    # quarter-integer data: exact +t / -t ties at the window edge, which the stable sort resolves by ROW ORDER, so
    # the order of the selection is visible in the result (defences.py:50, :70)
    g = (np.round(rng.standard_normal((n, cols)) * 64) / 64).astype(np.float32)
    order = rng.permutation(n)[:theta].astype(np.int32)
    c = 2 * int(n * MAL_PROP)
    want = faithful.trimmed_mean(g[order], theta, c)
    gd = eng.to_device(g)
    got = eng.trimmed_mean(gd, n, c, row_index=order).numpy()
    assert close(got, want)
    # continuous data as well
    g2 = rng.standard_normal((n, cols)).astype(np.float32)
    got2 = eng.trimmed_mean(eng.to_device(g2), n, c, row_index=order).numpy()
    assert close(got2, faithful.trimmed_mean(g2[order], theta, c))


def test_trimmed_mean_row_index_must_be_int32_on_the_device(eng):
    torch = pytest.importorskip('torch')
    g = torch.randn((50, 40), device='cuda')
    order = torch.randperm(50, device='cuda')[:31]           # int64, as torch.tensor(selection) would be
    want = ideal.trimmed_mean(g.cpu().numpy()[order.cpu().numpy()], 10)
    assert close(eng.trimmed_mean(g, 50, 10, row_index=order).cpu().numpy(), want)      # converted, not reinterpreted
    with pytest.raises(ValueError):
        eng.trimmed_mean(g, 50, 10, row_index=torch.tensor([0, 1, 50], device='cuda'))   # out of range
    with pytest.raises(ValueError):
        eng.trimmed_mean(g, 50, 10, row_index=order.cpu())                                 # wrong device


# ---- configs[4]'s flow at N = 10000 ----------------------------------------------------------------------
def test_config5_flow_n10000(eng):
    """attack -> ONE distance matrix -> Krum and Bulyan from it, N = 10000, f = m = 2400, on a D = 2048 slice."""
    torch = pytest.importorskip('torch')
    n, d = 10000, 2048
    f = int(n * MAL_PROP)
    g_host = scaled(4600, n, d)
    g = torch.from_numpy(g_host).cuda()
    drift, mean, std = eng.drift_attack(g[:f], 1.5, write_back=True)
    want_drift = faithful.drift_vector(g_host[:f].copy(), 1.5)
    assert np.array_equal(drift.cpu().numpy(), want_drift)      # m = 2400 rows: the reference's bits
    g_host[:f] = drift.cpu().numpy()
    assert torch.equal(g[:f], drift[None, :].expand(f, d))
    dist = eng.pairwise_distances(g)
    dist_host = dist.numpy()
    want64 = ideal.distance_matrix(g_host)
    mask = ~np.eye(n, dtype=bool)
    mask[:f, :f] = False
    assert np.all(dist_host[:f, :f][~np.eye(f, dtype=bool)] == 0.0)
    assert np.allclose(dist_host[mask], want64[mask], rtol=1e-6)
    assert np.array_equal(dist_host, dist_host.T)
    # the reference's decisions on this matrix
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist_host, n, f)
    sel = eng.bulyan_select(dist, n, f).tolist()
    check_selection(dist_host, n, f, sel)
    out = eng.trimmed_mean(g, n, 2 * f, row_index=np.asarray(sel, dtype=np.int32)).cpu().numpy()
    cols = np.random.default_rng(1).choice(d, 48, replace=False)
    assert close(out[cols], faithful.trimmed_mean(g_host[sel][:, cols], len(sel), 2 * f))
    # and the fused entry point agrees with the pieces
    out2, sel2 = eng.bulyan(g, n, f, return_selection=True)
    assert sel2.cpu().tolist() == sel and close(out2.cpu().numpy(), out)


# ---- the remaining exports ------------------------------------------------------------------
def test_host_convenience_entry_points(eng):
    import ctypes
    g = scaled(4700, 37, 900)
    n, f = 37, 8
    dist = np.empty((n, n), dtype=np.float32)
    rc = eng.lib.byz_pairwise_distances_host(eng.ctx, g.ctypes.data_as(ctypes.c_void_p), n, 900,
                                             dist.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    want = faithful.distance_matrix(g)
    off = ~np.eye(n, dtype=bool)
    assert np.allclose(dist[off], want[off], rtol=2e-6) and np.all(np.isinf(np.diag(dist)))
    idx = ctypes.c_int32(-7)
    assert eng.lib.byz_krum_select_host(eng.ctx, want.ctypes.data_as(ctypes.c_void_p), n, n, f, ctypes.byref(idx)) == 0
    assert idx.value == faithful.krum_pick(want, faithful.visit_order(n), n, f)
    eng.reserve(4000, 10000)                                   # byz_ctx_reserve: pre-sizes, must not disturb results
    assert eng.krum(g, n, f, return_index=True) == faithful.krum(g, n, f, return_index=True)


def test_drift_hook_called_directly(eng):
    """DriftAttack._attack_grads with arbitrary vectors (malicious.py:34-36) goes through byz_drift_axpy_dev."""
    from attacking_federate_learning_amd import malicious
    rng = np.random.default_rng(4800)
    mean, std = rng.standard_normal(70001).astype(np.float32), np.abs(rng.standard_normal(70001)).astype(np.float32)
    want = mean.copy()
    want[:] -= 1.5 * std[:]
    att = malicious.DriftAttack(1.5)
    got = att._attack_grads(mean, std, None, None)
    assert got is mean and np.allclose(got, want, rtol=1e-6, atol=1e-6)   # one fused multiply-add vs numpy's two roundings


def test_distances_to_dict_is_the_reference_dict(eng):
    g = scaled(4900, 9, 300)
    d = eng.pairwise_distances(g)
    as_dict = d.to_dict()
    assert list(as_dict.keys()) == [1, 0] + list(range(2, 9))
    dense = d.numpy()
    assert all(as_dict[i][j] == dense[i, j] for i in range(9) for j in range(9) if i != j)
    assert all(i not in as_dict[i] for i in range(9))


# ---- the ring selection of the trimmed mean (256 .. 2560 rows) and its hand-over to the general kernel --------------
@pytest.mark.parametrize('n,c', [(256, 60), (1000, 200), (1000, 0), (1001, 999), (1537, 300), (2080, 1920), (2560, 1000), (3000, 600), (5200, 4800)])
def test_ring_selection_resolves_continuous_columns(eng, n, c):
    """Continuous data: (nearly) every tile is resolved by the one-histogram ring selection, and the answer is the
    reference's.  A few columns per thousand have exact +t / -t ties at the window edge and go the general way."""
    torch = pytest.importorskip('torch')
    d = 4096
    g = scaled(5000 + n + c, n, d)
    gt = torch.from_numpy(g).cuda()
    got = eng.trimmed_mean(gt, n, c).cpu().numpy()
    redone = eng.trimmed_mean_redone()
    assert close(got, ideal.trimmed_mean(g, c))
    cols = np.random.default_rng(2).choice(d, 40, replace=False)
    assert close(got[cols], faithful.trimmed_mean(g[:, cols], n, c))
    assert redone <= max(2, (d // 16) // 20), 'ring selection handed %d of %d tiles back' % (redone, d // 16)


def test_ring_selection_hands_hard_tiles_to_the_general_kernel(eng):
    """Quantised columns (ties everywhere), a constant column, outliers and a NaN: all beyond the ring selection."""
    n, d = 1000, 640
    rng = np.random.default_rng(5100)
    g = rng.standard_normal((n, d)).astype(np.float32)
    g[:, 0:64] = np.round(g[:, 0:64] * 4) / 4           # exact ties at the window edge
    g[:, 64:80] = 1.25                                   # constant columns
    g[5, 100:140] = 1e30                                 # outliers squeeze everything into one bucket
    g[7, 200] = np.nan
    g[:300, 300:340] = g[0, 300:340]                     # 300 identical clients (the attack): more than one sort holds
    got = eng.trimmed_mean(g, n, 200)
    assert eng.trimmed_mean_redone() >= 10
    want = faithful.trimmed_mean(g, n, 200)
    assert np.isnan(got[200]) and np.isnan(want[200])
    ok = np.ones(d, dtype=bool)
    ok[200] = False
    assert close(got[ok], want[ok])


# ---- the pre-split (bf16 plane) Gram: the same arithmetic as the fused split kernel, bit for bit ---------------
@pytest.mark.parametrize('n,d,dup', [(2900, 3 * 8192 + 100, 0), (3000, 5 * 8192, 0), (3300, 2 * 8192 + 33 * 32 + 4, 700),
                                     (4000, 20 * 8192 + 36, 0)])
def test_plane_gram_is_bitwise_the_fused_gram(eng, monkeypatch, n, d, dup):
    """gram_planes.hip splits every operand once into bf16 planes (fragment order in HBM) and multiplies 256 x 128 tiles;
    gram.hip splits inside the 128 x 128 tile kernel.  Same six terms, same chain lengths, same slab order: the fp64 Gram
    must be IDENTICAL, with several super-chunks (a small plane budget), a ragged K tail, an odd number of slab rows,
    and through the identical-row shortcut's row indirection."""
    torch = pytest.importorskip('torch')
    gen = torch.Generator(device='cuda').manual_seed(900 + n)
    g = torch.randn((n, d), generator=gen, device='cuda', dtype=torch.float32)
    g *= (1.0 + 0.5 * torch.rand((n, 1), generator=gen, device='cuda'))
    if dup:
        g[torch.randperm(n, device='cuda')[:dup]] = g[7].clone()
    monkeypatch.setenv('BYZ_GRAM_MODE', 'split')          # bf16 x 3 on both sides (the planes' default is f16 x 2)
    monkeypatch.setenv('BYZ_GRAM_PLANES', '0')
    fused = eng.gram(g).clone()
    monkeypatch.setenv('BYZ_GRAM_PLANES', '1')
    one = eng.gram(g).clone()
    monkeypatch.setenv('BYZ_GRAM_PLANE_MB', '400')      # two chunks of planes at a time: several super-chunks
    many = eng.gram(g).clone()
    assert torch.equal(fused, one), float((fused - one).abs().max())
    assert torch.equal(fused, many), float((fused - many).abs().max())
    rows = [0, 1, n // 2, n - 1]
    host = g[rows].cpu().numpy().astype(np.float64)
    want = host @ host.T
    got = one[rows][:, rows].cpu().numpy()
    norms = np.sqrt(np.diag(want))
    worst = float(np.max(np.abs(got - want) / (norms[:, None] * norms[None, :])))
    assert worst < 1e-6, worst


@pytest.mark.parametrize('n,d,family', [(2900, 3 * 8192 + 100, 'scaled'), (4000, 6 * 8192 + 36, 'heavy'), (3000, 4 * 8192, 'ranges')])
def test_f16x2_gram_against_fp64(eng, monkeypatch, n, d, family):
    """The default long-K arithmetic: every value split into two fp16 planes (per row and 8192-column chunk scaled by a
    power of two), three MFMAs per block.  Against fp64 on sampled rows: c_ij within 2e-7 |g_i| |g_j| (the reference's
    own sdot is 1e-6 .. 1e-4 there); identical rows give exact zero distances and identical distance rows; the result
    does not depend on how the columns are cut into super-chunks; a second call reproduces it bit for bit."""
    torch = pytest.importorskip('torch')
    gen = torch.Generator(device='cuda').manual_seed(1700 + n)
    g = torch.randn((n, d), generator=gen, device='cuda', dtype=torch.float32)
    if family == 'scaled':
        g *= (1.0 + 0.5 * torch.rand((n, 1), generator=gen, device='cuda'))
    elif family == 'heavy':      # a few huge coordinates per row: the scale follows the chunk's largest magnitude
        g *= torch.where(torch.rand((n, d), generator=gen, device='cuda') < 1e-3, 1.0e3, 1.0)
    else:                        # rows and column ranges of wildly different magnitude, one all-zero row
        g *= torch.pow(10.0, torch.randint(-12, 12, (n, 1), generator=gen, device='cuda').float())
        g[:, 8192:2 * 8192] *= 1.0e-6
        g[11] = 0.0
    g[5] = g[n - 1]
    g[1700] = g[5]
    monkeypatch.delenv('BYZ_GRAM_MODE', raising=False)
    monkeypatch.setenv('BYZ_GRAM_PLANES', '1')
    one = eng.gram(g).clone()
    monkeypatch.setenv('BYZ_GRAM_PLANE_MB', '300')
    many = eng.gram(g).clone()
    monkeypatch.delenv('BYZ_GRAM_PLANE_MB')
    assert torch.equal(one, many)
    assert torch.equal(one, eng.gram(g))
    rows = [0, 1, 5, 11, 1700, n // 2, n - 2, n - 1]
    host = g[rows].cpu().numpy().astype(np.float64)
    want = host @ host.T
    got = one[rows][:, rows].cpu().numpy()
    norms = np.sqrt(np.diag(want))
    scale = np.maximum(norms[:, None] * norms[None, :], np.finfo(np.float64).tiny)
    worst = float(np.max(np.abs(got - want) / scale))
    assert worst < 2e-7, worst
    dist = eng.pairwise_distances(g).numpy()
    assert dist[5, 1700] == 0.0 and dist[5, n - 1] == 0.0 and dist[1700, n - 1] == 0.0
    keep = np.ones(n, dtype=bool)
    keep[[5, 1700, n - 1]] = False
    assert np.array_equal(dist[5, keep], dist[1700, keep]) and np.array_equal(dist[5, keep], dist[n - 1, keep])
    if family != 'ranges':       # distances against fp64 on the sampled rows (1e-6, the float bar of north_star)
        sq = np.diag(want)
        d_want = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * want, 0.0))
        d_got = dist[np.ix_(rows, rows)]
        off = ~np.eye(len(rows), dtype=bool) & (d_want > 0)
        assert float(np.max(np.abs(d_got[off] - d_want[off]) / d_want[off])) < 1e-6


# ---- written at the end of round 2 without a GPU at hand: kept LAST, so that a surprise here cannot hide the results above ----
def gaussian(seed, n, d):
    return np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)


def test_trimmed_mean_lds_kernel_returns_nan_for_a_column_with_nan(eng):
    """Above 5632 rows the LDS bitonic kernel takes over; a NaN anywhere in a column makes np.median -- and with it the
    reference's result -- NaN there, as it does below 5632 rows, and leaves the other columns alone."""
    n, d, c = 6000, 10, 100
    g = gaussian(61, n, d)
    g[17, 3] = np.nan
    g[5999, 9] = np.nan
    got = np.asarray(eng.trimmed_mean(g, n, c))
    assert np.isnan(got[3]) and np.isnan(got[9])
    clean = [0, 1, 2, 4, 5, 6, 7, 8]
    assert close(got[clean], ideal.trimmed_mean(g[:, clean], c))


def test_reserve_covers_the_small_path(eng):
    """byz_ctx_reserve for N <= 128 pre-sizes csrc/krum_small.hip's workspaces; results are what they were."""
    g = scaled(77, 100, 21840)
    before = eng.krum(g, 100, 24, return_index=True)
    eng.reserve(100, 79510)
    assert eng.krum(g, 100, 24, return_index=True) == before
    assert np.array_equal(eng.krum(g, 100, 24), g[before])


def test_near_duplicate_rows_agree_to_rounding_with_and_without_the_identical_row_shortcut(eng):
    """VERDICT r1 (weak 3): the pair (near-duplicate row, group of identical rows) is re-evaluated on the difference itself
    (defences.py:20) in both paths, so the two distance matrices agree to fp32 rounding everywhere -- no 2e-2 allowance."""
    import os
    n, d = 700, 40000
    rng = np.random.default_rng(123)
    g = scaled(124, n, d)
    group = np.sort(rng.choice(n, size=300, replace=False))
    g[group] = g[group[0]]
    near = int(np.setdiff1d(np.arange(n), group)[17])
    g[near] = g[group[0]]
    g[near, 3000] += 1.0
    with_shortcut = eng.pairwise_distances(g).numpy()
    os.environ['BYZ_GRAM_DEDUP'] = '0'
    try:
        without = eng.pairwise_distances(g).numpy()
    finally:
        del os.environ['BYZ_GRAM_DEDUP']
    off = ~np.eye(n, dtype=bool)
    true_d = abs(float(g[near, 3000]) - float(g[group[0], 3000]))
    assert abs(with_shortcut[near, group[0]] - true_d) < 1e-5 and abs(without[near, group[0]] - true_d) < 1e-5
    assert np.allclose(with_shortcut[off], without[off], rtol=1e-5, atol=1e-6)


def test_small_krum_loop_form_for_long_rows(eng, monkeypatch):
    """N <= 128 with more than 98,304 columns runs the general path by default; with the limit raised, K1's LOOP form (more
    than eight slices per workgroup) must give the same distances and the same index."""
    n, d, f = 100, 300001, 24
    g = scaled(4321, n, d)
    base_d = eng.pairwise_distances(g).numpy()
    base_i = eng.krum(g, n, f, return_index=True)
    monkeypatch.setenv('BYZ_KRUM_SMALL_MAX_COLS', str(1 << 20))
    loop_d = eng.pairwise_distances(g).numpy()
    loop_i = eng.krum(g, n, f, return_index=True)
    eng.check()
    off = ~np.eye(n, dtype=bool)
    assert np.all(np.isinf(np.diag(loop_d))) and np.array_equal(loop_d, loop_d.T)
    assert np.allclose(loop_d[off], base_d[off], rtol=1e-6, atol=0.0)
    assert loop_i == base_i == faithful.krum_pick(loop_d, faithful.visit_order(n), n, f)


@pytest.mark.parametrize('d', [117706, 200000, 262144])
def test_small_krum_with_four_to_eight_slices_per_workgroup(eng, monkeypatch, d):
    """D = 117,706 is the reference's Cifar10Net (data_sets.py:33-52): four 128-column slices per workgroup; 200,000 and
    2^18 take seven and eight.  Same distances and index as the general path."""
    n, f = 100, 24
    g = scaled(5000 + d % 1000, n, d)
    monkeypatch.setenv('BYZ_KRUM_SMALL_MAX_COLS', str(1 << 18))   # the default stops at three slices per workgroup
    small_d = eng.pairwise_distances(g).numpy()
    small_i = eng.krum(g, n, f, return_index=True)
    monkeypatch.setenv('BYZ_KRUM_SMALL', '0')
    base_d = eng.pairwise_distances(g).numpy()
    base_i = eng.krum(g, n, f, return_index=True)
    eng.check()
    off = ~np.eye(n, dtype=bool)
    assert np.array_equal(small_d, small_d.T)
    assert np.allclose(small_d[off], base_d[off], rtol=1e-6, atol=0.0)
    assert small_i == base_i == faithful.krum_pick(small_d, faithful.visit_order(n), n, f)

