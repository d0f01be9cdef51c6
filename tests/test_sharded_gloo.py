"""The N>1 orchestration (attacking_federate_learning_amd.sharded) on CPU: world_size 2 over gloo.

The per-rank kernels are replaced by a numpy stand-in built on the oracle (TEST ONLY: the package itself has
no CPU implementation); what is under test is the sharding plan, the exchange step and the replicated
selection -- sharded results must equal the unsharded oracle's.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import faithful, ideal


class OracleKernels:
    """numpy stand-in for HipKernels (same method names); tensors are CPU torch tensors."""

    def gram(self, g_local):
        g = g_local.numpy().astype(np.float64)
        return torch.from_numpy(g @ g.T)

    def gram_share(self, panel, row_index, share_count, share_index):
        """numpy stand-in of byz_gram_share_dev: every share_count-th 128 x 128 lower-triangle tile, in list order."""
        rows = panel.numpy().astype(np.float64)
        if row_index is not None:
            rows = rows[np.asarray(row_index)]
        full = rows @ rows.T
        n = len(rows)
        out = np.zeros_like(full)
        tile, position = 4, 0       # tiny tiles so that a 23-row test really splits the work
        for ti in range(0, n, tile):
            for tj in range(0, ti + 1, tile):
                if position % share_count == share_index:
                    out[ti:ti + tile, tj:tj + tile] = full[ti:ti + tile, tj:tj + tile]
                    out[tj:tj + tile, ti:ti + tile] = full[tj:tj + tile, ti:ti + tile]
                position += 1
        return torch.from_numpy(out)

    def gram_accumulator(self, n, like):
        return torch.zeros((n, n), dtype=torch.float64)

    def gram_share_add(self, panel, row_index, share_count, share_index, gram):
        """byz_gram_share_add_dev: the share is added into the caller's buffer, other ranks' tiles untouched."""
        gram.add_(self.gram_share(panel, row_index, share_count, share_index))
        return gram

    # ---- near-duplicate pairs: the same protocol as libbyzagg (gram.hip): distances_from_gram lists every pair i > j with
    # d^2 < (c_ii + c_jj) / 16 in CANONICAL order (ascending i, then j -- a function of the all-reduced Gram alone, so
    # slot p means the same pair on every rank), and POISONS their Gram-identity distances, so that a result is only right
    # if the count -> per-rank sums of squared differences -> all-reduce -> apply path really ran across the ranks.
    def __init__(self):
        self.pairs = []

    def near_pairs_count(self):
        return len(self.pairs)

    def near_pairs_sqdist(self, panel, count, row_index=None):
        rows = panel.numpy()
        if row_index is not None:
            rows = rows[np.asarray(row_index)]
        assert count == len(self.pairs)
        sq = np.zeros(count, dtype=np.float64)
        for p, (i, j) in enumerate(self.pairs):
            diff = (rows[i] - rows[j]).astype(np.float32).astype(np.float64)      # fp32 difference (defences.py:20)
            sq[p] = float((diff * diff).sum())
        return torch.from_numpy(sq)

    def near_pairs_apply(self, sq, dist):
        sq = sq.numpy()
        for p, (i, j) in enumerate(self.pairs):
            dist[i, j] = dist[j, i] = np.float32(np.sqrt(sq[p]))

    def distances_from_gram(self, gram, local_columns=None, all_reduce=None):
        c = gram.numpy()
        sq = np.diag(c)
        d2 = np.maximum(sq[:, None] + sq[None, :] - 2 * c, 0.0)
        d2 = np.minimum(d2, d2.T)
        d = np.sqrt(d2).astype(np.float32)
        np.fill_diagonal(d, np.inf)
        n = len(d)
        self.pairs = [(i, j) for i in range(n) for j in range(i) if d2[i, j] < (sq[i] + sq[j]) / 16.0]
        for i, j in self.pairs:
            d[i, j] = d[j, i] = np.nan
        if local_columns is not None and self.pairs:
            part = self.near_pairs_sqdist(local_columns, len(self.pairs))
            if all_reduce is not None:
                all_reduce(part)
            self.near_pairs_apply(part, d)
        return d

    def krum_select(self, d, users_count, corrupted_count):
        return faithful.krum_pick(d, faithful.visit_order(len(d)), users_count, corrupted_count)

    def bulyan_select(self, d, users_count, corrupted_count, on_device=False):
        return ideal.bulyan_selection(d, users_count, corrupted_count)

    def trimmed_mean(self, g_local, corrupted_count, row_index=None):
        g = g_local.numpy()
        if row_index is not None:
            g = g[np.asarray(row_index)]
        return torch.from_numpy(faithful.trimmed_mean(g, len(g), corrupted_count))

    def no_defense(self, g_local):
        return torch.from_numpy(np.mean(g_local.numpy(), axis=0))

    def drift(self, rows_local, num_std, write_back=False):
        rows = rows_local.numpy()
        mean, std = faithful.attack_statistics(rows)
        vec = mean - np.float32(num_std) * std
        if write_back:
            rows_local[:] = torch.from_numpy(vec)
        return torch.from_numpy(vec), torch.from_numpy(mean), torch.from_numpy(std)

    def column_chain(self, rows_local, carry=None, mean=None):
        """One link of numpy's chain (oracle.faithful.attack_statistics_sequential cut at an owner boundary)."""
        a = rows_local.numpy()
        s = np.zeros(a.shape[1], dtype=np.float32) if carry is None else carry.numpy().copy()
        mu = None if mean is None else mean.numpy()
        for r in range(a.shape[0]):
            if mu is None:
                s = s + a[r]
            else:
                dlt = a[r] - mu
                s = s + dlt * dlt
        return torch.from_numpy(s)

    def column_finish(self, total_rows, num_std, sum=None, sumsq=None, mean=None):
        m = np.float32(total_rows)
        if sum is not None:
            return torch.from_numpy(sum.numpy() / m)
        std = np.sqrt(sumsq.numpy() / m)
        return torch.from_numpy(std), torch.from_numpy(mean.numpy() - np.float32(num_std) * std)

    def row(self, g_local, index):
        return g_local[index].clone()


def make_matrix(n, d, f, seed=5):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    # two honest clients that nearly coincide with two others (beyond the malicious rows 0..f-1): pairs the Gram identity
    # cannot resolve, which the ranks re-evaluate on the difference itself and sum through the small all-reduce
    g[7] = g[6] + np.float32(1e-3) * rng.standard_normal(d).astype(np.float32)
    g[11] = g[10] + np.float32(2e-3) * rng.standard_normal(d).astype(np.float32)
    return g


def worker(rank, world, port, n, d, f, results):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from attacking_federate_learning_amd.sharded import ShardedAggregator
        agg = ShardedAggregator(OracleKernels())
        g = make_matrix(n, d, f)
        # start client-sharded (uneven on purpose), convert with the exchange step
        rows_per_rank = [n // world + (1 if r < n % world else 0) for r in range(world)]
        start = sum(rows_per_rank[:rank])
        mine = torch.from_numpy(g[start:start + rows_per_rank[rank]].copy())
        g_local = agg.reshard_clients_to_columns(mine, rows_per_rank)
        lo, hi = agg.column_slices(d)[rank]
        assert np.array_equal(g_local.numpy(), g[:, lo:hi])
        out = {}
        # attack first (in place on the local slice), then the defences on the attacked matrix
        drift, _, _ = agg.drift_attack(g_local, f, 1.5, write_back=True, gather=True)
        out['drift'] = drift.numpy()
        out['attacked_slice'] = g_local.numpy().copy()
        out['krum_index'] = agg.krum(g_local, n, f, return_index=True)
        out['krum'] = agg.krum(g_local, n, f, gather=True).numpy()
        out['tm'] = agg.trimmed_mean(g_local, n, f, gather=True).numpy()
        out['nodef'] = agg.no_defense(g_local, gather=True).numpy()
        b, sel = agg.bulyan(g_local, n, f, gather=True, return_selection=True, total_columns=d)
        out['bulyan'], out['selection'] = b.numpy(), np.asarray(sel)
        # ---- the clients layout (north_star): rows stay where the clients left them
        mine2 = torch.from_numpy(g[start:start + rows_per_rank[rank]].copy())
        drift2, _, _ = agg.drift_attack_clients(mine2, rows_per_rank, f, 1.5)
        out['c_drift'] = drift2.numpy()
        out['c_rows'] = mine2.numpy().copy()
        out['c_krum_index'] = agg.krum_clients(mine2, rows_per_rank, n, f, return_index=True)
        out['c_krum'] = agg.krum_clients(mine2, rows_per_rank, n, f).numpy()
        # small panels on purpose: several gathers, a ragged last one
        out['c_dist'] = np.asarray(agg.client_distances(mine2, rows_per_rank, panel_columns=50))
        b2, sel2 = agg.bulyan_clients(mine2, rows_per_rank, n, f, return_selection=True)
        out['c_bulyan'], out['c_selection'] = b2.numpy(), np.asarray(sel2)
        out['comm'] = agg.comm_report()
        results[rank] = out
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world,n,d,f', [(2, 23, 157, 5), (2, 12, 64, 2), (4, 23, 157, 5), (3, 12, 64, 2)])
def test_ranks_equal_the_unsharded_oracle(world, n, d, f):
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(worker, args=(world, free_port(), n, d, f, results), nprocs=world, join=True)
        results = dict(results)
    g = make_matrix(n, d, f)
    g[:f] = faithful.drift_vector(g[:f].copy(), 1.5)
    want_bulyan, want_sel = faithful.bulyan(g, n, f, return_selection=True)
    for rank in range(world):
        r = results[rank]
        assert np.array_equal(r['drift'], g[0])
        assert r['krum_index'] == faithful.krum(g, n, f, return_index=True)
        assert np.array_equal(r['krum'], g[r['krum_index']])
        assert np.array_equal(r['tm'], faithful.trimmed_mean(g, n, f))
        assert np.array_equal(r['nodef'], faithful.no_defense(g, n, f))
        assert r['selection'].tolist() == want_sel
        assert np.array_equal(r['bulyan'], want_bulyan)
        # clients layout: same answers, full vectors on every rank
        # the attack's statistics are ONE chain of additions handed from rank to rank: the reference's bits, not 1e-6
        assert np.array_equal(r['c_drift'], g[0])
        lo = sum(n // world + (1 if q < n % world else 0) for q in range(rank))
        assert np.array_equal(r['c_rows'], g[lo:lo + len(r['c_rows'])])
        assert r['c_krum_index'] == r['krum_index']
        assert np.allclose(r['c_krum'], g[r['krum_index']], rtol=1e-6, atol=1e-6)
        want_dist = ideal.distance_matrix(g)
        off = ~np.eye(n, dtype=bool)
        assert np.allclose(r['c_dist'][off], want_dist[off], rtol=1e-5)
        assert r['c_selection'].tolist() == want_sel
        assert np.allclose(r['c_bulyan'], want_bulyan, rtol=1e-5, atol=1e-6)
        if world > 1:
            assert r['comm']['allgather_row_tiles']['calls'] >= 2 and r['comm']['allgather_row_tiles']['bytes'] > 0
            assert 'reshard_selected_rows' in r['comm'] and 'allreduce_gram' in r['comm']
            # the near-duplicate pairs (two planted ones + the attack's identical rows) went through the exchange, in both
            # layouts (the stub poisons their Gram-identity distances: the results above are right only if they did)
            assert r['comm']['allreduce_near_pairs']['calls'] >= 2


def subgroup_worker(rank, world, port, members, n, d, m, f, results):
    """A process group that does NOT start at global rank 0: every send / recv / broadcast of the sharded layer must
    address peers by GLOBAL rank (drift_attack_clients passed group-local ranks)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        group = dist.new_group(ranks=members)        # collective over the whole world: every rank calls it
        if rank not in members:
            return
        from attacking_federate_learning_amd.sharded import ShardedAggregator
        agg = ShardedAggregator(OracleKernels(), group=group)
        assert agg.world == len(members) and agg.rank == members.index(rank)
        g = make_matrix(n, d, m)
        w = agg.world
        rows_per_rank = [n // w + (1 if r < n % w else 0) for r in range(w)]
        start = sum(rows_per_rank[:agg.rank])
        mine = torch.from_numpy(g[start:start + rows_per_rank[agg.rank]].copy())
        out = {}
        drift, _, _ = agg.drift_attack_clients(mine, rows_per_rank, m, 1.5)
        out['c_drift'] = drift.numpy()
        out['c_rows'] = mine.numpy().copy()
        out['c_krum_index'] = agg.krum_clients(mine, rows_per_rank, n, f, return_index=True)
        out['c_krum'] = agg.krum_clients(mine, rows_per_rank, n, f).numpy()
        b, sel = agg.bulyan_clients(mine, rows_per_rank, n, f, return_selection=True)
        out['c_bulyan'], out['c_selection'] = b.numpy(), np.asarray(sel)
        # ... and the columns layout through the same group
        mine3 = torch.from_numpy(g[start:start + rows_per_rank[agg.rank]].copy())
        g_local = agg.reshard_clients_to_columns(mine3, rows_per_rank)
        agg.drift_attack(g_local, m, 1.5, write_back=True)
        out['krum_index'] = agg.krum(g_local, n, f, return_index=True)
        b2, sel2 = agg.bulyan(g_local, n, f, gather=True, return_selection=True, total_columns=d)
        out['bulyan'], out['selection'] = b2.numpy(), np.asarray(sel2)
        results[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,members,n,d,m,f', [(3, [1, 2], 23, 157, 14, 5), (4, [1, 3], 12, 64, 7, 2)])
def test_a_subgroup_that_does_not_start_at_global_rank_zero(world, members, n, d, m, f):
    """m malicious rows reach into the SECOND member's rows, so the chain hop (recv / send) and both broadcasts of
    drift_attack_clients run between global ranks (1, 2) / (1, 3) while the group-local ranks are (0, 1)."""
    with mp.Manager() as manager:
        results = manager.dict()
        mp.spawn(subgroup_worker, args=(world, free_port(), members, n, d, m, f, results), nprocs=world, join=True)
        results = dict(results)
    assert sorted(results) == members
    g = make_matrix(n, d, m)
    g[:m] = faithful.drift_vector(g[:m].copy(), 1.5)
    want_bulyan, want_sel = faithful.bulyan(g, n, f, return_selection=True)
    want_krum = faithful.krum(g, n, f, return_index=True)
    w = len(members)
    for rank in members:
        r = results[rank]
        assert np.array_equal(r['c_drift'], g[0])
        lo = sum(n // w + (1 if q < n % w else 0) for q in range(members.index(rank)))
        assert np.array_equal(r['c_rows'], g[lo:lo + len(r['c_rows'])])
        assert r['c_krum_index'] == want_krum and r['krum_index'] == want_krum
        assert np.allclose(r['c_krum'], g[want_krum], rtol=1e-6, atol=1e-6)
        assert r['c_selection'].tolist() == want_sel and r['selection'].tolist() == want_sel
        assert np.allclose(r['c_bulyan'], want_bulyan, rtol=1e-5, atol=1e-6)
        assert np.array_equal(r['bulyan'], want_bulyan)


def test_column_slices_cover_everything():
    from attacking_federate_learning_amd.sharded import ShardedAggregator

    class Fake(ShardedAggregator):
        def __init__(self, world):
            self.world, self.rank = world, 0

    for world in (1, 2, 3, 8):
        for d in (1, 7, 8, 79510):
            b = Fake(world).column_slices(d)
            assert b[0][0] == 0 and b[-1][1] == d
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
