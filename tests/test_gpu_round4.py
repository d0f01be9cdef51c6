"""Round-4 regression tests of the HIP path (needs an MI355X): the cases ADVICE r3 found by reading the code."""
import numpy as np
import pytest

from oracle import faithful, scale

pytestmark = pytest.mark.gpu


def point_distances(seed, n, dim=16):
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((n, dim)).astype(np.float64)
    sq = (pts * pts).sum(1)
    dist = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (pts @ pts.T), 0.0)).astype(np.float32)
    dist = np.minimum(dist, dist.T)
    np.fill_diagonal(dist, np.inf)
    return dist


@pytest.mark.parametrize('n', [40, 129, 300, 1000])
@pytest.mark.parametrize('slack', ['zero', 'beyond'])
def test_krum_with_an_empty_prefix_is_the_reference(eng, n, slack):
    """users_count - corrupted_count == 0 (or <= -(n - 1)): sorted(...)[:k] is empty, sum([]) == 0 < 1e20 for every row and the
    first row visited -- row 1 -- wins (defences.py:27-37).  Reachable through return_index=True, which skips the assert.  The
    general path's row sort used to skip the store of the empty sum and krum_argmin read stale scores."""
    dist = point_distances(n, n)
    users, corrupted = (n, n) if slack == 'zero' else (n, 2 * n + 5)
    want = faithful.krum_pick(dist, faithful.visit_order(n), users, corrupted)
    assert want == 1
    # poison the context's score buffer with a run whose scores are large and whose winner is not row 1
    eng.krum_select(dist[::-1, ::-1].copy(), n, 1)
    assert eng.krum_select(dist, users, corrupted) == want
    g = np.random.default_rng(n).standard_normal((n, 64)).astype(np.float32)
    from attacking_federate_learning_amd import defences
    assert defences.krum(g, users, corrupted, return_index=True) == faithful.krum(g, users, corrupted, return_index=True) == 1


def test_engine_krum_takes_the_reference_positional_debug(eng):
    """krum(users_grads, users_count, corrupted_count, distances, return_index, debug): `debug` sits where the engine's `check`
    keyword used to be: a positional True must not switch the safety check off, nor raise."""
    torch = pytest.importorskip('torch')
    g = torch.randn((12, 300), device='cuda')
    idx = eng.krum(g, 12, 3, None, True, True)
    row = eng.krum(g, 12, 3, None, False, True)
    assert torch.equal(row, g[idx])
    with pytest.raises(TypeError):
        eng.krum(g, 12, 3, None, False, True, False)


@pytest.mark.parametrize('n,twins', [(2944, 100), (3000, 40)])
def test_twins_in_row_blocks_split_at_different_scales(eng, monkeypatch, n, twins):
    """The attack's identical rows, too few of them to drop a row of Gram tiles (so the Gram runs over all N rows), spread over
    many 32-row blocks of the operand split, with a wide dynamic range inside every row (elements far below 2^-17 of the
    row's largest: their fp16 planes are subnormal and depend on the shift) and planted outliers that send SOME blocks of the
    sampled split to the exact two-pass redo: twins then sit in blocks split at different shifts and their Gram entries are
    no longer bitwise equal.  Their distance must still be exactly zero and their distance rows identical -- by
    the byte-for-byte proof of dedup.hip, not by cancellation -- and the selections the reference's on those distances."""
    torch = pytest.importorskip('torch')
    d = 3 * 8192 + 64
    gen = torch.Generator(device='cuda').manual_seed(n)
    g = torch.randn((n, d), generator=gen, device='cuda', dtype=torch.float32)
    g *= torch.pow(2.0, torch.randint(-30, 1, (n, d), generator=gen, device='cuda').float())   # 30 binades inside a row
    rows = torch.randperm(n, generator=gen, device='cuda')[:twins]
    twin = g[rows[0]].clone()
    g[rows] = twin
    # outliers 2^12 above anything sampled, in unsampled columns, in a few row blocks that also hold a twin
    for r in rows[: twins // 3].tolist():
        other = (r // 32) * 32 + ((r + 1) % 32)
        if other < n and other not in rows.tolist():
            g[other, 5 + 8192] = 4096.0
            g[other, 777] = -8192.0
    monkeypatch.delenv('BYZ_GRAM_MODE', raising=False)
    dist = eng.pairwise_distances(g).numpy()
    idx = sorted(rows.tolist())
    sub = dist[np.ix_(idx, idx)]
    assert np.all(sub[~np.eye(len(idx), dtype=bool)] == 0.0)
    keep = np.ones(n, dtype=bool)
    keep[idx] = False
    for r in idx[1:]:
        assert np.array_equal(dist[idx[0], keep], dist[r, keep])
    assert np.array_equal(dist, dist.T)
    # against the exact two-pass split: the same matrix to rounding
    monkeypatch.setenv('BYZ_GRAM_SPLIT_TWO_PASS', '1')
    exact = eng.pairwise_distances(g).numpy()
    monkeypatch.delenv('BYZ_GRAM_SPLIT_TWO_PASS')
    off = ~np.eye(n, dtype=bool)
    assert np.allclose(dist[off], exact[off], rtol=1e-6, atol=0.0)
    assert np.array_equal(dist[off] == 0.0, exact[off] == 0.0)
    # sampled fp64 check of non-twin distances
    some = [0, 1, idx[0], n // 2, n - 1]
    host = g[some].cpu().numpy().astype(np.float64)
    for a in range(len(some)):
        for b in range(a):
            want = np.sqrt(((host[a].astype(np.float32) - host[b].astype(np.float32)).astype(np.float64) ** 2).sum())
            assert abs(dist[some[a], some[b]] - want) <= 1e-6 * want
    f = int(0.24 * n)
    assert eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)
    got = list(eng.bulyan_select(dist, n, f))
    assert got == scale.bulyan_selection(dist, n, f)


@pytest.mark.parametrize('n,d,dup', [(2900, 3 * 8192 + 100, 0), (3300, 2 * 8192 + 33 * 32 + 4, 700), (4000, 5 * 8192 + 36, 0),
                                     (2817, 2 * 8192 + 7, 0), (3009, 8192 + 640, 0)])
def test_gram_with_skipped_blocks_is_bitwise_the_slab_granular_gram(eng, monkeypatch, n, d, dup):
    """Round 5: gram_planes_kernel no longer multiplies 32 x 32 blocks nobody reads (rows past the matrix, blocks strictly above
    the diagonal of a diagonal slab).  Every block that IS computed goes through the same MFMA chain in the
    same order as under round 4's slab-granular rule (BYZ_GRAM_BLOCK_SKIP=0): the fp64 Gram must be IDENTICAL -- with a ragged K
    tail, row counts that leave 1 / 2 / 3 / 4 valid row blocks in the last slab, several super-chunks and the identical-row
    indirection -- and equal to fp64 on sampled entries."""
    torch = pytest.importorskip('torch')
    gen = torch.Generator(device='cuda').manual_seed(4100 + n)
    g = torch.randn((n, d), generator=gen, device='cuda', dtype=torch.float32)
    g *= (1.0 + 0.5 * torch.rand((n, 1), generator=gen, device='cuda'))
    if dup:
        g[torch.randperm(n, device='cuda')[:dup]] = g[7].clone()
    monkeypatch.delenv('BYZ_GRAM_MODE', raising=False)
    monkeypatch.setenv('BYZ_GRAM_PLANES', '1')
    monkeypatch.setenv('BYZ_GRAM_BLOCK_SKIP', '0')
    base = eng.gram(g).clone()
    monkeypatch.setenv('BYZ_GRAM_BLOCK_SKIP', '1')
    lean = eng.gram(g).clone()
    monkeypatch.setenv('BYZ_GRAM_PLANE_MB', '300')
    lean_many = eng.gram(g).clone()
    eng.check()
    assert torch.equal(base, lean), float((base - lean).abs().max())
    assert torch.equal(base, lean_many)
    assert torch.equal(lean, lean.T)
    # round 5, second change: the chunks' level-1 sums are written out and added into the fp64 slabs by a kernel behind the
    # tile kernel (chunk order, the same fp64 operations) instead of read-modify-written under a ticket inside it
    # (BYZ_GRAM_DEFER=0: the in-kernel update): bitwise the same, with one super-chunk and with several, skip on and off
    monkeypatch.setenv('BYZ_GRAM_DEFER', '0')
    inline_many = eng.gram(g).clone()
    monkeypatch.delenv('BYZ_GRAM_PLANE_MB')
    inline_one = eng.gram(g).clone()
    monkeypatch.setenv('BYZ_GRAM_BLOCK_SKIP', '0')
    inline_all_blocks = eng.gram(g).clone()
    monkeypatch.setenv('BYZ_GRAM_DEFER', '1')
    deferred_all_blocks = eng.gram(g).clone()
    eng.check()
    assert torch.equal(inline_many, lean) and torch.equal(inline_one, lean)
    assert torch.equal(inline_all_blocks, lean) and torch.equal(deferred_all_blocks, lean)
    # round 6: a workgroup of the deferred kernel walks a SPAN of consecutive chunks without leaving (BYZ_GRAM_KSPAN; the DMA
    # ring runs across the chunk boundaries, each chunk's level-1 sums leave at its boundary): bitwise the same for every span --
    # spans that divide the chunk count and spans that do not, one longer than the matrix, with one super-chunk and several
    monkeypatch.setenv('BYZ_GRAM_BLOCK_SKIP', '1')
    for span, plane_mb in ((2, None), (3, None), (64, None), (2, '300'), (5, '300')):
        monkeypatch.setenv('BYZ_GRAM_KSPAN', str(span))
        if plane_mb:
            monkeypatch.setenv('BYZ_GRAM_PLANE_MB', plane_mb)
        spanned = eng.gram(g).clone()
        eng.check()
        monkeypatch.delenv('BYZ_GRAM_PLANE_MB', raising=False)
        assert torch.equal(spanned, lean), (span, plane_mb, float((spanned - lean).abs().max()))
    monkeypatch.delenv('BYZ_GRAM_KSPAN')
    # round 6: the tile list in bands with rounds of consecutive (chunk, tile) units dealt to the XCDs in turn (the default)
    # against rounds 2-5's super-block list with a contiguous share per XCD (BYZ_GRAM_ORDER=0) -- which workgroup computes which
    # (tile, chunk) changes, nothing else: bitwise; also with spans, no round gate, the in-kernel update and several super-chunks
    # (BYZ_GRAM_CLAIM=0: the XCDs take their runs of units in turn instead of claiming them from one counter)
    for env in ({}, {'BYZ_GRAM_KSPAN': '3'}, {'BYZ_GRAM_ROUND': '0'}, {'BYZ_GRAM_DEFER': '0', 'BYZ_GRAM_PLANE_MB': '300'},
                {'BYZ_GRAM_ROUND': '5', 'BYZ_GRAM_PLANE_MB': '300'}, {'BYZ_GRAM_CLAIM': '0'},
                {'BYZ_GRAM_CLAIM': '0', 'BYZ_GRAM_DEFER': '0', 'BYZ_GRAM_PLANE_MB': '300'},
                {'BYZ_GRAM_CLAIM': '0', 'BYZ_GRAM_ROUND': '0', 'BYZ_GRAM_KSPAN': '2'}):
        for order in ('0', '1'):
            monkeypatch.setenv('BYZ_GRAM_ORDER', order)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            ordered = eng.gram(g).clone()
            eng.check()
            for k in env:
                monkeypatch.delenv(k)
            assert torch.equal(ordered, lean), (order, env, float((ordered - lean).abs().max()))
    monkeypatch.delenv('BYZ_GRAM_ORDER')
    # the last rows and the diagonal against fp64
    rows = torch.tensor([0, 1, 31, 32, 63, 64, 127, 128, n - 33, n - 32, n - 2, n - 1], device='cuda')
    want = g[rows].double() @ g.double().T
    got = lean[rows]
    scale = (g.double() ** 2).sum(1).sqrt()
    # 12 x N entries: the largest error seen is 4.05e-7 of |g_i| |g_j| at K = 16,391, the shortest K this path takes (the
    # 8 x 8 samples of test_f16x2_gram_against_fp64 stay below 2e-7 from K = 24,676); the distances' bar is 1e-6
    assert float(((got - want).abs() / (scale[rows][:, None] * scale[None, :])).max()) < 1e-6


def test_collect_gradients_writes_all_device_clients_in_one_launch(eng, golden):
    """server.py:81-83 over clients whose gradients are per-parameter device tensors: `byz_assemble_rows_dev` places all of
    them with ONE launch (pointer table on the device).  Bit-exact against the reference's golden assembly and against the
    oracle's row assembly on awkward sizes and alignments; host-vector clients in between still take the row copy."""
    torch = pytest.importorskip('torch')
    from attacking_federate_learning_amd.assembly import GradientMatrix

    class Client:
        def __init__(self, grads):
            self.grads = grads

    c = golden['assemble_4x204']
    gm = GradientMatrix(4, 204, engine=eng)
    gm.collect_gradients([Client([eng.to_device(c['u%d_t%d' % (u, t)]) for t in range(5)]) for u in range(4)])
    eng.synchronize()
    assert np.array_equal(gm.numpy(), c['G'])

    rng = np.random.default_rng(13)
    sizes = [1, 3, 784 * 100, 100, 7, 100 * 10, 10, 5, 4096, 33] * 4          # 40 tensors per client
    d = sum(sizes)
    n = 9
    gm = GradientMatrix(n, d, engine=eng, torch_device='cuda')
    want = np.empty((n, d), dtype=np.float32)
    users = []
    for u in range(n):
        pool = torch.from_numpy(rng.standard_normal(d + 64).astype(np.float32)).cuda()
        tensors, off = [], u % 5          # misaligned views of one pool: every alignment case of the copy
        for k in sizes:
            tensors.append(pool[off:off + k])
            off += k
        faithful.assemble_row(want, u, [t.cpu().numpy() for t in tensors])
        # clients 3 and 4 still hand over the reference's flat host vector (user.py:92): the run of device clients is cut
        users.append(Client(want[u].copy() if u in (3, 4) else tensors))
    eng.timing(True)
    gm.collect_gradients(users)
    torch.cuda.synchronize()
    launches = eng.timing_read().get('misc', {}).get('launches', 0)
    eng.timing(False)
    assert np.array_equal(gm.numpy(), want)
    assert launches == 2, launches        # rows 0-2 and rows 5-8: one launch each, whatever the number of clients
    with pytest.raises(ValueError):
        eng.assemble_rows(gm.data, 0, [users[0].grads, users[1].grads[:-1]])
    with pytest.raises(ValueError):
        eng.assemble_rows(gm.data, 7, [users[0].grads] * 3)      # rows 7..9 of a 9-row matrix


def test_collect_gradients_reuses_the_pointer_table_while_no_tensor_moved(eng):
    """Round 5: the table of the clients' tensor addresses goes to the device when an address changed, not
    every round.  Second round, same tensors, NEW contents: the rows are the new contents (byz_assemble_rows_again_dev, no
    upload).  Then one client's tensor is replaced by another: the table is rebuilt.  Then a second matrix with other clients
    in between: the key follows the context's one table."""
    torch = pytest.importorskip('torch')
    import ctypes
    from attacking_federate_learning_amd.assembly import GradientMatrix

    class Client:
        def __init__(self, grads):
            self.grads = grads

    shapes = [(100, 784), (100,), (10, 100), (10,)]        # MnistNet (data_sets.py:13-23)
    d = sum(int(np.prod(sh)) for sh in shapes)
    n = 17
    gen = torch.Generator(device='cuda').manual_seed(77)
    users = [Client([torch.randn(sh, device='cuda', generator=gen) for sh in shapes]) for _ in range(n)]

    def want():
        return torch.stack([torch.cat([t.reshape(-1) for t in u.grads]) for u in users]).cpu().numpy()

    gm = GradientMatrix(n, d, engine=eng, torch_device='cuda')
    gm.collect_gradients(users)
    assert np.array_equal(gm.numpy(), want()) and eng._assemble_key is not None
    first_key = eng._assemble_key
    for u in users:                      # a new round: the same .grad buffers, new values
        for t in u.grads:
            t.normal_(generator=gen)
    gm.data.zero_()
    eng.timing(True)
    gm.collect_gradients(users)
    torch.cuda.synchronize()
    assert eng.timing_read().get('misc', {}).get('launches', 0) == 1
    eng.timing(False)
    assert np.array_equal(gm.numpy(), want()) and eng._assemble_key is first_key
    users[5].grads[2] = torch.randn(shapes[2], device='cuda', generator=gen)     # one tensor moved
    gm.collect_gradients(users)
    assert np.array_equal(gm.numpy(), want()) and eng._assemble_key != first_key
    # another matrix, other clients: the context holds ONE table, the key says whose
    other = [Client([torch.randn(sh, device='cuda', generator=gen) for sh in shapes]) for _ in range(n)]
    gm2 = GradientMatrix(n, d, engine=eng, torch_device='cuda')
    gm2.collect_gradients(other)
    gm.data.zero_()
    gm.collect_gradients(users)
    assert np.array_equal(gm.numpy(), want())
    # a non-contiguous tensor is copied to a temporary: nothing to reuse next round
    users[0].grads[0] = torch.randn((784, 100), device='cuda', generator=gen).t()
    gm.collect_gradients(users)
    assert np.array_equal(gm.numpy(), want()) and eng._assemble_key is None
    # the raw entry point refuses a shape the device table does not hold
    rc = eng.lib.byz_assemble_rows_again_dev(eng.ctx, ctypes.c_void_p(gm.data.data_ptr()), n, d, d, 0, n - 1, len(shapes), None)
    assert rc == -1


@pytest.mark.parametrize('n,identical', [(40, 0), (200, 0), (700, 168), (3000, 0), (4000, 960)])
def test_krum_and_bulyan_from_one_row_sort(eng, n, identical):
    """byz_krum_bulyan_select_dev: Krum's index and Bulyan's selection from one distance matrix with ONE sort of its rows
    (BASELINE configs[4] runs both on the same distances): the values the two separate calls return, and the C oracle's."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_scale import check_selection, point_distances as scale_points
    f = int(0.24 * n)
    dist = scale_points(5100 + n, n, identical=identical)
    idx, sel = eng.krum_bulyan_select(dist, n, f)
    assert idx == eng.krum_select(dist, n, f) == scale.krum_pick(dist, n, f)
    assert list(sel) == list(eng.bulyan_select(dist, n, f))
    check_selection(dist, n, f, list(sel))
    idx2, sel_dev = eng.krum_bulyan_select(eng.pairwise_distances(np.random.default_rng(n).standard_normal((n, 64)).astype(np.float32)), n, f, on_device=True)
    assert len(sel_dev.numpy()) == n - 2 * f and 0 <= idx2 < n
