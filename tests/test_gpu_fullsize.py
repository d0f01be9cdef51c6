"""The headline configurations at their FULL size, checked through sampled values (needs an MI355X with ~170 GB free).

BENCH's default workload is BASELINE.json configs[3]: Bulyan over N = 4000 clients x D = 10,000,000 parameters
(reference defences.py:55-70) -- 160 GB of gradients, 1221 chunks of 8192 columns through the fp16 x 2 plane Gram in ten
super-chunks, the slab/ticket accumulation, a 2080-row second stage.  No oracle finishes that size, so the check goes
through what CAN be recomputed independently:

  * distances: 8 sampled rows of the matrix against a literal restatement of defences.py:20 evaluated on the device in
    column chunks -- the fp32 difference, its squares and their sum in fp64 (elementwise torch ops: no fp64 matmul, no
    Gram identity) -- to 1e-6 relative;
  * selection: given the engine's OWN distance matrix the C oracle (oracle/scale.py, the reference's loop) must reproduce
    the selection pick for pick (N = 4000: the whole loop; N = 10,000: every 40th pick and both ends);
  * output: 64 sampled columns of the 2080 (5200) selected rows, gathered to the host, through oracle.faithful.trimmed_mean
    (defences.py:44-52) to 1e-5.

The same for the slice of configs[4] one GPU of eight holds (N = 10,000, D = 3,125,000; attack + Krum + Bulyan, 125 GB).
"""
import numpy as np
import pytest

from oracle import faithful, scale

pytestmark = pytest.mark.gpu

MAL_PROP = 0.24


def make_matrix(torch, n, d, seed, device):
    """bench.py's generator ('scaled' family): row blocks on the device, no second copy."""
    gen = torch.Generator(device=device).manual_seed(seed)
    g = torch.empty((n, d), dtype=torch.float32, device=device)
    perm = torch.from_numpy(np.random.default_rng(seed).permutation(n)).to(device)
    scale_ = (1.0 + 0.5 * perm.to(torch.float32) / n)
    rows_per = max(1, (1 << 28) // max(d, 1))
    for r in range(0, n, rows_per):
        blk = g[r:r + rows_per]
        blk.normal_(generator=gen)
        blk.mul_(scale_[r:r + rows_per, None])
    return g


def literal_distances(torch, g, rows, chunk=1 << 17):
    """defences.py:20 for the sampled rows against every row: norm of the fp32 difference, squares summed in fp64."""
    n, d = g.shape
    acc = torch.zeros((len(rows), n), dtype=torch.float64, device=g.device)
    for lo in range(0, d, chunk):
        block = g[:, lo:lo + chunk]
        for k, r in enumerate(rows):
            diff = (block - block[r]).double()          # the difference is formed in fp32, as the reference does
            acc[k] += (diff * diff).sum(dim=1)
            del diff
    return acc.sqrt().cpu().numpy()


def free_memory_gb(torch):
    free, _ = torch.cuda.mem_get_info()
    return free / 2 ** 30


def check_sampled(torch, eng, g, n, f, dist, selection, out, label):
    theta = n - 2 * f
    rng = np.random.default_rng(11)
    # ---- distances of 8 sampled rows (two of them among the first f: the malicious rows when the attack ran)
    rows = sorted(set([0, f // 2] + rng.choice(n, 6, replace=False).tolist()))
    want = literal_distances(torch, g, rows)
    got = dist[rows].astype(np.float64)
    for k, r in enumerate(rows):
        want[k, r] = np.inf
    finite = np.isfinite(want) & (want > 0)
    rel = np.abs(got[finite] - want[finite]) / want[finite]
    assert rel.max() < 1e-6, '%s: sampled distances off by %.2e relative' % (label, rel.max())
    assert np.array_equal(got == 0, want == 0), '%s: exact zeros between identical rows' % label
    # ---- the selection is the reference's on the engine's own distances
    sel = [int(s) for s in selection]
    assert len(sel) == theta and len(set(sel)) == theta
    if n <= 4000:
        ref_sel = scale.bulyan_selection(dist, n, f)
        assert sel == ref_sel, '%s: first difference at pick %d' % (label, next(i for i, (a, b) in enumerate(zip(sel, ref_sel)) if a != b))
    else:
        picks = np.unique(np.concatenate([np.arange(0, theta, 40), np.arange(12), np.arange(theta - 12, theta)])).astype(np.int32)
        bad, first, expected = scale.verify_picks(dist, n, f, sel, picks)
        assert bad == 0, '%s: pick %d: reference picks row %d, got %d' % (label, first, expected, sel[first])
    # ---- 64 sampled output columns through the reference's trimmed mean on the gathered selected rows
    d = g.shape[1]
    cols = np.sort(rng.choice(d, 64, replace=False))
    cols[0], cols[-1] = 0, d - 1
    sub = g[:, torch.as_tensor(cols, device=g.device)][torch.as_tensor(sel, device=g.device)].cpu().numpy()   # columns first: (n, 64)
    want_out = faithful.trimmed_mean(sub, theta, 2 * f)
    got_out = out[torch.as_tensor(cols, device=out.device)].cpu().numpy()
    assert np.allclose(got_out, want_out, rtol=1e-5, atol=1e-5), '%s: output columns off by %.3e' % (
        label, np.abs(got_out - want_out).max())
    assert bool(torch.isfinite(out).all())


def test_config4_full_size_sampled(eng):
    import torch
    device = torch.device('cuda', eng.device)
    torch.cuda.empty_cache()
    if free_memory_gb(torch) < 200:
        pytest.skip('needs ~200 GB of free device memory')
    n, d = 4000, 10_000_000
    f = int(n * MAL_PROP)
    g = make_matrix(torch, n, d, 1237, device)
    dist = eng.pairwise_distances(g).numpy()
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    torch.cuda.synchronize()
    eng.check()
    check_sampled(torch, eng, g, n, f, dist, sel.cpu().numpy(), out, 'c4')
    del g
    torch.cuda.empty_cache()


def test_config5_slice_full_size_sampled(eng):
    """One GPU's slice of configs[4]: drift attack over the first f rows written back (malicious.py:10-36), Krum and Bulyan
    on one distance matrix (N = 10,000, D = 3,125,000; the attack makes 2400 rows one vector: exact ties everywhere)."""
    import torch
    device = torch.device('cuda', eng.device)
    torch.cuda.empty_cache()
    if free_memory_gb(torch) < 180:
        pytest.skip('needs ~180 GB of free device memory')
    n, d = 10000, 3_125_000
    f = int(n * MAL_PROP)
    g = make_matrix(torch, n, d, 1237, device)
    # the attack's statistics on 4096 sampled columns (the first and the last tile of the register-resident kernel among them),
    # before the rows are overwritten: numpy's BITS at the full height of 2400 rows and the full width of the slice
    picked = np.random.default_rng(12).choice(d, 4096 - 96, replace=False)
    cols_host = np.unique(np.concatenate([picked, np.arange(64), np.arange(d - 32, d)]))
    cols = torch.as_tensor(cols_host, device=device)
    head = g[:f][:, cols].cpu().numpy()
    drift, mean, std = eng.drift_attack(g[:f], 1.5, write_back=True)
    want_mean, want_std = faithful.attack_statistics(head)
    want_drift = faithful.drift_vector(head.copy(), 1.5)
    assert np.array_equal(mean[cols].cpu().numpy(), want_mean) and np.array_equal(std[cols].cpu().numpy(), want_std)
    assert np.array_equal(drift[cols].cpu().numpy(), want_drift)
    assert bool((g[f - 1][cols] == drift[cols]).all()) and bool((g[0][cols] == drift[cols]).all())
    cols = cols[:64]      # (the sampled output columns further down)
    dist_dev = eng.pairwise_distances(g)
    dist = dist_dev.numpy()
    idx = eng.krum_select(dist_dev, n, f)
    assert idx == scale.krum_pick(dist, n, f)
    out, sel = eng.bulyan(g, n, f, return_selection=True)
    torch.cuda.synchronize()
    eng.check()
    check_sampled(torch, eng, g, n, f, dist, sel.cpu().numpy(), out, 'c5 slice')
    del g
    torch.cuda.empty_cache()
