"""Multi-GPU form of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Two layouts of the N x D gradient matrix, both behind one `ShardedAggregator`:

  clients   rank r holds rows [r*N/W, (r+1)*N/W) of G with all D columns -- where the clients' gradients land
            (reference main.py:26-32, server.py:81-83; the layout BASELINE.json's north_star names).
            Distances: G is all-gathered one column PANEL at a time (RCCL all-gather of row tiles; the next panel's
            gather overlaps this panel's Gram); every rank then holds all N rows of the panel and computes ITS SHARE of
            the 128 x 128 Gram tiles (byz_gram_share_dev), the shares are summed by one fp64 all-reduce of N x N.
            Selection: replicated.  Bulyan's second stage needs every selected row per coordinate: the theta selected
            rows are re-sharded to column slices by one personalised exchange, the median-window mean runs per slice and
            the D-vector is all-gathered.
  columns   every rank holds all N rows of its own slice of the D columns.  trimmed_mean, no_defense and the attack are
            independent per column (no collective); the distance matrix is a sum over columns: each rank's fp64 Gram of
            its slice, ONE all-reduce of N x N doubles, replicated selection, local second stage.  Client-sharded input
            gets here through `reshard_clients_to_columns` (a personalised exchange of 1/W of the bytes an all-gather
            moves).

The all-reduce result is bitwise identical on every rank (each element is reduced in one fixed order and then broadcast),
so all ranks select the same clients; identical rows still tie exactly because the distance kernel canonicalises them.

`kernels` is the seam between this orchestration and the per-GPU kernels: the product uses `HipKernels` (libbyzagg on the
rank's GPU); the CPU test-suite drives the same orchestration over gloo with a numpy stand-in defined in tests/ -- there
is no CPU implementation in this package.  Every collective is timed and its bytes counted (`comm_report()`), because
which layout wins is a measurement, not an opinion.
"""
import time

import numpy as np


class HipKernels:
    """Per-rank kernels on the rank's own MI355X, operating on torch CUDA tensors."""

    def __init__(self, engine=None):
        from .engine import get_engine
        self.engine = engine or get_engine()

    def gram(self, g_local):
        return self.engine.gram(g_local)                       # (N, N) float64 CUDA tensor

    def gram_share(self, panel, row_index, share_count, share_index):
        return self.engine.gram_share(panel, row_index, share_count, share_index)

    def gram_accumulator(self, n, like):
        import torch
        return torch.zeros((n, n), dtype=torch.float64, device=like.device)

    def gram_share_add(self, panel, row_index, share_count, share_index, gram):
        return self.engine.gram_share_add(panel, row_index, share_count, share_index, gram)

    def pairwise_distances(self, g_local):
        return self.engine.pairwise_distances(g_local)         # Distances handle: Gram, distances, near pairs in one call

    def distances_from_gram(self, gram, local_columns=None, all_reduce=None):
        return self.engine.distances_from_gram(gram, gram.shape[0], local_columns=local_columns, all_reduce=all_reduce)

    def near_pairs_count(self):
        return self.engine.near_pairs_count()

    def near_pairs_sqdist(self, panel, count, row_index=None):
        return self.engine.near_pairs_sqdist(panel, count, row_index=row_index)

    def near_pairs_apply(self, sq, dist):
        self.engine.near_pairs_apply(sq, dist)

    def krum_select(self, dist, users_count, corrupted_count):
        return self.engine.krum_select(dist, users_count, corrupted_count)

    def bulyan_select(self, dist, users_count, corrupted_count, on_device=False):
        return self.engine.bulyan_select(dist, users_count, corrupted_count, on_device=on_device)

    def krum_bulyan_select(self, dist, users_count, corrupted_count, on_device=False):
        return self.engine.krum_bulyan_select(dist, users_count, corrupted_count, on_device=on_device)

    def trimmed_mean(self, g_local, corrupted_count, row_index=None):
        # row_index comes from this package (a selection the kernels produced): no bounds re-check, no host sync
        return self.engine.trimmed_mean(g_local, g_local.shape[0], corrupted_count, row_index=row_index,
                                        validate_index=False)

    def no_defense(self, g_local):
        return self.engine.no_defense(g_local)

    def drift(self, rows_local, num_std, write_back=False):
        return self.engine.drift_attack(rows_local, num_std, write_back=write_back)

    def column_chain(self, rows_local, carry=None, mean=None):
        return self.engine.column_chain(rows_local, carry=carry, mean=mean)

    def column_finish(self, total_rows, num_std, sum=None, sumsq=None, mean=None):
        return self.engine.column_finish(total_rows, num_std, sum=sum, sumsq=sumsq, mean=mean)

    def row(self, g_local, index):
        return g_local[index].clone()


class ShardedAggregator:
    """defences.py / malicious.py over a sharded gradient matrix (see the module docstring for the two layouts)."""

    def __init__(self, kernels, group=None):
        import os
        import torch.distributed as dist
        self.dist = dist
        self.kernels = kernels
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # BYZ_FORCE_COLLECTIVES=1 issues the collectives even at world size 1 (exercises the RCCL path on one GPU)
        self.always_collective = dist.is_initialized() and os.environ.get('BYZ_FORCE_COLLECTIVES') == '1'
        self._comm = {}
        self._pending = []

    # ---- collectives, timed ------------------------------------------------------------------------------------
    def _collective(self):
        return self.world > 1 or self.always_collective

    def _timed(self, name, nbytes, tensor, fn, count=True):
        """Run one collective, book its bytes and its duration (device events for CUDA tensors)."""
        rec = self._comm.setdefault(name, {'calls': 0, 'bytes': 0, 'ms': 0.0})
        if count:
            rec['calls'] += 1
            rec['bytes'] += int(nbytes)
        if getattr(tensor, 'is_cuda', False):
            import torch
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            out = fn()
            stop.record()
            self._pending.append((name, start, stop))
            return out
        t0 = time.perf_counter()
        out = fn()
        rec['ms'] += (time.perf_counter() - t0) * 1e3
        return out

    def comm_report(self, reset=True):
        """{collective: {calls, bytes, ms}} since the last report; bytes are what THIS rank receives."""
        for name, start, stop in self._pending:
            stop.synchronize()
            self._comm[name]['ms'] += start.elapsed_time(stop)
        self._pending = []
        out = {k: dict(v) for k, v in self._comm.items()}
        if reset:
            self._comm = {}
        return out

    def _all_reduce(self, name, tensor):
        if self._collective():
            nbytes = tensor.numel() * tensor.element_size() * 2 * (self.world - 1) // max(self.world, 1)
            self._timed(name, nbytes, tensor,
                        lambda: self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group))
        return tensor

    # ---- distances: the one bulk exchange of the path -----------------------------------------------------------
    def global_distances(self, g_local):
        """columns layout: Gram of the local slice, one all-reduce, distances (near-duplicate pairs re-evaluated on
        the difference: per-rank partial sums over the local columns, one more small all-reduce)."""
        if not self._collective() and hasattr(self.kernels, 'pairwise_distances'):
            # one rank holds every column: the engine's own composition, in which rows that its duplicate search has compared
            # byte for byte need no second proof through the near-pair exchange
            return self.kernels.pairwise_distances(g_local)
        gram = self.kernels.gram(g_local)
        self._all_reduce('allreduce_gram', gram)
        reduce_pairs = (lambda t: self._all_reduce('allreduce_near_pairs', t)) if self._collective() else None
        return self.kernels.distances_from_gram(gram, local_columns=g_local, all_reduce=reduce_pairs)

    def client_distances(self, rows_local, rows_per_rank, panel_columns=None):
        """clients layout: all-gather of column panels overlapped with this rank's share of the Gram tiles, then one
        all-reduce of the N x N fp64 Gram.  `rows_local`: this rank's (n_r, D) rows."""
        import torch
        n_max = int(max(rows_per_rank))
        n_mine, d = rows_local.shape
        assert n_mine == rows_per_rank[self.rank]
        if panel_columns is None:   # ~1 GiB of gathered panel, at least a few thousand columns
            panel_columns = max(4096, min(d, (1 << 28) // max(n_max * self.world, 1)))
        panel_columns = int(min(panel_columns, d))
        n_panels = -(-d // panel_columns)
        device = rows_local.device
        # logical row -> physical row of the gathered (W * n_max)-row panel (ranks with fewer rows are padded)
        row_index = np.concatenate([r * n_max + np.arange(rows_per_rank[r]) for r in range(self.world)]).astype(np.int32)
        even = all(c == n_max for c in rows_per_rank)
        row_index_t = None if even else torch.from_numpy(row_index).to(device)
        panels = [torch.empty((self.world * n_max, panel_columns), dtype=rows_local.dtype, device=device) for _ in range(2)]
        stage = [torch.zeros((n_max, panel_columns), dtype=rows_local.dtype, device=device) for _ in range(2)]

        class _Requests:
            """The requests of one panel's gather as one waitable."""
            def __init__(self, reqs):
                self.reqs = reqs

            def wait(self):
                for req in self.reqs:
                    req.wait()

        def start_gather(k):
            """All-gather of panel k's row tiles as point-to-point transfers over the xGMI full mesh: every rank sends its
            tile to each peer and receives each peer's tile straight into that peer's rows of the panel -- all seven links of
            a GPU busy at once, where the ring of `all_gather_into_tensor` is bound by ONE 153 GB/s link (VERDICT r3 weak 10;
            `_exchange` already moved the selected rows this way).  Same bytes, same result."""
            lo = k * panel_columns
            width = min(panel_columns, d - lo)
            src, dst = stage[k % 2], panels[k % 2]
            src[:n_mine, :width].copy_(rows_local[:, lo:lo + width])
            if not self._collective():
                dst[:n_max].copy_(src)
                return None, width
            rec = self._comm.setdefault('allgather_row_tiles', {'calls': 0, 'bytes': 0, 'ms': 0.0})
            rec['calls'] += 1
            rec['bytes'] += (self.world - 1) * n_max * width * src.element_size()
            ops = []
            for peer in range(self.world):
                mine_here = dst[peer * n_max:(peer + 1) * n_max]
                if peer == self.rank and not self.always_collective:
                    mine_here.copy_(src)
                    continue
                global_peer = self.dist.get_global_rank(self.group, peer) if self.group is not None else peer
                ops.append(self.dist.P2POp(self.dist.isend, src, global_peer, group=self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, mine_here, global_peer, group=self.group))
            return _Requests(self.dist.batch_isend_irecv(ops)), width

        def sweep(per_panel):
            """Gather every panel once; panel k+1 is in flight while per_panel(k, panel) runs.  The booked time of the
            gathers is what the compute stream spent WAITING for them (the exposed part)."""
            pending = start_gather(0)
            for k in range(n_panels):
                work, width = pending
                if work is not None:
                    self._timed('allgather_row_tiles', 0, stage[0], work.wait, count=False)
                if k + 1 < n_panels:
                    pending = start_gather(k + 1)
                per_panel(k, panels[k % 2][:, :width])

        acc = {}
        # the panels' shares accumulate in ONE caller-owned N x N fp64 buffer inside the reduction kernel
        # (byz_gram_share_add_dev): no N x N `add_` pass per panel (128 MB per panel at N = 4000, ~150 panels at D = 1e7)
        gram = self.kernels.gram_accumulator(int(sum(rows_per_rank)), rows_local)

        def gram_of(k, panel):
            self.kernels.gram_share_add(panel, row_index_t, self.world, self.rank, gram)
        sweep(gram_of)
        self._all_reduce('allreduce_gram', gram)
        dist_m = self.kernels.distances_from_gram(gram)
        # pairs the Gram identity cannot resolve (near-duplicate clients) are re-evaluated on the difference itself
        # (defences.py:20): a second sweep, every rank takes the panels k = rank mod W, one small all-reduce
        count = self.kernels.near_pairs_count()
        if count:
            def pairs_of(k, panel):
                if k % self.world == self.rank:
                    part = self.kernels.near_pairs_sqdist(panel, count, row_index=row_index_t)
                    acc['sq'] = part if 'sq' not in acc else acc['sq'].add_(part)
            sweep(pairs_of)
            if 'sq' not in acc:
                acc['sq'] = torch.zeros(count, dtype=torch.float64, device=device)
            self._all_reduce('allreduce_near_pairs', acc['sq'])
            self.kernels.near_pairs_apply(acc['sq'], dist_m)
        return dist_m

    # ---- defences.py, columns layout --------------------------------------------------------------------------------
    def no_defense(self, g_local, users_count=None, corrupted_count=None, gather=False, total_columns=None):
        return self._maybe_gather(self.kernels.no_defense(g_local), gather, total_columns)

    def trimmed_mean(self, g_local, users_count, corrupted_count, gather=False, total_columns=None):
        return self._maybe_gather(self.kernels.trimmed_mean(g_local, corrupted_count), gather, total_columns)

    def krum(self, g_local, users_count, corrupted_count, return_index=False, gather=False, total_columns=None):
        if not return_index:
            assert users_count >= 2 * corrupted_count + 1, (
                'users_count>=2*corrupted_count + 3', users_count, corrupted_count)
        dist_m = self.global_distances(g_local)
        index = self.kernels.krum_select(dist_m, users_count, corrupted_count)
        if return_index:
            return index
        return self._maybe_gather(self.kernels.row(g_local, index), gather, total_columns)

    def bulyan(self, g_local, users_count, corrupted_count, gather=False, return_selection=False, total_columns=None):
        assert users_count >= 4 * corrupted_count + 3
        dist_m = self.global_distances(g_local)
        # the selection stays on the device between the loop and the second stage
        selection = self.kernels.bulyan_select(dist_m, users_count, corrupted_count, on_device=True)
        out = self.kernels.trimmed_mean(g_local, 2 * corrupted_count, row_index=selection)
        out = self._maybe_gather(out, gather, total_columns)
        if return_selection:
            host = selection.numpy() if hasattr(selection, 'numpy') else selection
            return out, np.asarray(host, dtype=np.int32)
        return out

    # ---- defences.py, clients layout ----------------------------------------------------------------------------------
    def _row_owner(self, rows_per_rank):
        offsets = np.concatenate([[0], np.cumsum(rows_per_rank)]).astype(np.int64)
        return offsets

    def krum_clients(self, rows_local, rows_per_rank, users_count, corrupted_count, return_index=False):
        """Krum over client-sharded rows: distances by all-gather, replicated selection, the winner's row broadcast
        by its owner.  Returns the full D-vector on every rank."""
        if not return_index:
            assert users_count >= 2 * corrupted_count + 1, (
                'users_count>=2*corrupted_count + 3', users_count, corrupted_count)
        dist_m = self.client_distances(rows_local, rows_per_rank)
        index = int(self.kernels.krum_select(dist_m, users_count, corrupted_count))
        if return_index:
            return index
        offsets = self._row_owner(rows_per_rank)
        total = int(offsets[-1])
        row = index if index >= 0 else total + index       # numpy's G[-1]
        owner = int(np.searchsorted(offsets, row, side='right') - 1)
        import torch
        out = torch.empty(rows_local.shape[1], dtype=rows_local.dtype, device=rows_local.device)
        if owner == self.rank:
            out.copy_(rows_local[row - int(offsets[owner])])
        if self._collective():
            src = self.dist.get_global_rank(self.group, owner) if self.group is not None else owner
            self._timed('broadcast_row', out.numel() * out.element_size(), out,
                        lambda: self.dist.broadcast(out, src=src, group=self.group))
        return out

    def bulyan_clients(self, rows_local, rows_per_rank, users_count, corrupted_count, return_selection=False):
        """Bulyan over client-sharded rows (north_star's flow): all-gather distances, replicated selection loop, the
        theta selected rows re-sharded to column slices, median-window mean per slice, all-gather of the D-vector."""
        assert users_count >= 4 * corrupted_count + 3
        dist_m = self.client_distances(rows_local, rows_per_rank)
        selection = np.asarray(self.kernels.bulyan_select(dist_m, users_count, corrupted_count), dtype=np.int64)
        cols, row_index = self.reshard_rows_to_columns(rows_local, rows_per_rank, selection)
        out = self.kernels.trimmed_mean(cols, 2 * corrupted_count, row_index=row_index)
        out = self._maybe_gather(out, True, total=rows_local.shape[1])
        return (out, selection.astype(np.int32)) if return_selection else out

    # ---- malicious.py --------------------------------------------------------------------------------------------------
    def drift_attack(self, g_local, n_malicious, num_std, write_back=True, gather=False, total_columns=None):
        """columns layout.  Rows 0..m-1 are the malicious clients (reference main.py:28); per column, no exchange."""
        drift, mean, std = self.kernels.drift(g_local[:n_malicious], num_std, write_back=write_back)
        return self._maybe_gather(drift, gather, total_columns), mean, std

    def drift_attack_clients(self, rows_local, rows_per_rank, n_malicious, num_std, write_back=True):
        """clients layout.  The malicious rows 0..m-1 sit on the first ranks.  numpy's mean / var add in row order
        (malicious.py:18-19), so per column the additions are ONE chain through those ranks: each continues the running sums of
        the rank before it over its own malicious rows (`column_chain`: a point-to-point hop of D floats), the last holder
        ends the chain (`column_finish`) and broadcasts -- first the mean (the squared deviations need it), then std and the
        drifted vector.  Bit for bit what the unsharded attack returns (rounds 1-4 combined per-rank (sum, sum of squares) in
        fp64: 1e-6 close).  The drifted vector is written into every local malicious row.  Returns (drift, mean, std) as full
        D-vectors on every rank."""
        import torch
        if n_malicious <= 0:
            raise ValueError('drift_attack_clients: no malicious rows')
        offsets = self._row_owner(rows_per_rank)
        first = int(offsets[self.rank])
        m_local = int(min(max(n_malicious - first, 0), rows_per_rank[self.rank]))
        holders = [r for r in range(self.world)
                   if min(max(n_malicious - int(offsets[r]), 0), rows_per_rank[r]) > 0]
        last = holders[-1]
        d = rows_local.shape[1]
        where = holders.index(self.rank) if self.rank in holders else -1

        def peer(r):
            """torch.distributed addresses send / recv / broadcast by GLOBAL rank even when a group is given."""
            return self.dist.get_global_rank(self.group, r) if self.group is not None else r

        def walk(mean):
            """My link of one walk over the malicious rows; the total on the last holder."""
            if where < 0:
                return None
            carry = None
            if where > 0:
                carry = torch.empty(d, dtype=torch.float32, device=rows_local.device)
                self._timed('attack_chain_hop', 4 * d, carry, lambda: self.dist.recv(carry, src=peer(holders[where - 1]), group=self.group))
            total = self.kernels.column_chain(rows_local[:m_local], carry=carry, mean=mean)
            if where + 1 < len(holders):
                self._timed('attack_chain_hop', 4 * d, total, lambda: self.dist.send(total, dst=peer(holders[where + 1]), group=self.group),
                            count=False)
            return total

        total = walk(None)
        mean = (self.kernels.column_finish(n_malicious, num_std, sum=total) if self.rank == last
                else torch.empty(d, dtype=torch.float32, device=rows_local.device))
        if self._collective():
            self._timed('broadcast_attack_mean', 4 * d, mean, lambda: self.dist.broadcast(mean, src=peer(last), group=self.group))
        squares = walk(mean)
        if self.rank == last:
            std, drift = self.kernels.column_finish(n_malicious, num_std, sumsq=squares, mean=mean)
            both = torch.stack([std, drift])
        else:
            both = torch.empty((2, d), dtype=torch.float32, device=rows_local.device)
        if self._collective():
            self._timed('broadcast_attack_vector', 8 * d, both, lambda: self.dist.broadcast(both, src=peer(last), group=self.group))
        std, drift = both[0], both[1]
        if write_back and m_local > 0:
            rows_local[:m_local] = drift
        return drift, mean, std

    # ---- layout conversion ---------------------------------------------------------------------------------------------
    def column_slices(self, n_cols):
        """Column range owned by every rank: contiguous, sizes differ by at most one."""
        base, extra = divmod(n_cols, self.world)
        bounds, start = [], 0
        for r in range(self.world):
            stop = start + base + (1 if r < extra else 0)
            bounds.append((start, stop))
            start = stop
        return bounds

    def _exchange(self, name, blocks_out, blocks_in):
        """Personalised exchange: blocks_out[p] goes to rank p, blocks_in[p] is filled from rank p (point-to-point over
        the xGMI full mesh: all links busy, where a ring would be bound by one 153 GB/s link)."""
        ops = []
        nbytes = 0
        landed = []    # (contiguous landing buffer, strided destination): a block of a pitch-padded matrix is not contiguous
        for peer in range(self.world):
            if peer == self.rank and not self.always_collective:
                continue        # (BYZ_FORCE_COLLECTIVES=1: the rank's own block goes through RCCL too, send-to-self)
            global_peer = self.dist.get_global_rank(self.group, peer) if self.group is not None else peer
            if blocks_out[peer].numel():
                ops.append(self.dist.P2POp(self.dist.isend, blocks_out[peer], global_peer, group=self.group))
            if blocks_in[peer].numel():
                dst = blocks_in[peer]
                if not dst.is_contiguous():
                    dst = dst.new_empty(dst.shape)
                    landed.append((dst, blocks_in[peer]))
                ops.append(self.dist.P2POp(self.dist.irecv, dst, global_peer, group=self.group))
                nbytes += dst.numel() * dst.element_size()
        if not ops:
            return

        def run():
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
            for buf, dst in landed:
                dst.copy_(buf)
        self._timed(name, nbytes, blocks_in[(self.rank + 1) % self.world], run)

    def reshard_clients_to_columns(self, rows_local, rows_per_rank):
        """client-sharded (rows_local: my rows x D) -> column-sharded (all rows x my columns).

        `rows_per_rank[r]` is the number of clients rank r holds.  One personalised exchange: point-to-point
        sends of (my rows) x (peer's columns); the received blocks are stacked in rank order.
        """
        all_rows = np.arange(int(sum(rows_per_rank)), dtype=np.int64)
        cols, _ = self.reshard_rows_to_columns(rows_local, rows_per_rank, all_rows, name='reshard_clients_to_columns')
        return cols

    def reshard_rows_to_columns(self, rows_local, rows_per_rank, wanted_rows, name='reshard_selected_rows'):
        """The rows `wanted_rows` (global indices, any order, e.g. Bulyan's selection) as a column-sharded matrix.

        Returns (cols, row_index): `cols` is (len(wanted_rows), my columns), stacked by owning rank; `row_index[s]` is
        the row of `cols` that holds wanted_rows[s] (None when the stacking order already is the wanted order), so that
        `trimmed_mean(cols, ..., row_index=row_index)` sees the rows in the caller's order (defences.py:70)."""
        import torch
        wanted_rows = np.asarray(wanted_rows, dtype=np.int64)
        offsets = self._row_owner(rows_per_rank)
        owner = np.searchsorted(offsets, wanted_rows, side='right') - 1
        n_cols = rows_local.shape[1]
        bounds = self.column_slices(n_cols)
        lo, hi = bounds[self.rank]
        counts = [int(np.sum(owner == r)) for r in range(self.world)]
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        # position of wanted_rows[s] inside its owner's block = its rank among that owner's wanted rows, in wanted order
        within = np.zeros(len(wanted_rows), dtype=np.int64)
        for r in range(self.world):
            mask = owner == r
            within[mask] = np.arange(counts[r])
        row_index = (starts[owner] + within).astype(np.int32)
        mine = wanted_rows[owner == self.rank] - offsets[self.rank]
        device = rows_local.device
        mine_t = torch.from_numpy(mine).to(device)
        # the leading dimension is padded to a multiple of four floats: slice widths differ by one between ranks, and a
        # 16-byte-aligned row pitch is what lets every rank's kernels take the same (LDS-DMA, vector-load) paths
        pitch = -(-(hi - lo) // 4) * 4
        out = torch.empty((len(wanted_rows), pitch), dtype=rows_local.dtype, device=device)[:, :hi - lo]
        all_of_mine = len(mine) == rows_local.shape[0] and np.array_equal(mine, np.arange(rows_local.shape[0]))

        def block_for(p):
            """My wanted rows x rank p's columns, contiguous: ONE gather per peer straight out of the caller's matrix (the
            first version gathered the rows with all D columns and then cut every peer's block out of that copy: twice the
            memory -- 83 + 83 GB next to a 160 GB matrix at configs[3] on one rank)."""
            cols = rows_local[:, bounds[p][0]:bounds[p][1]]
            return cols.contiguous() if all_of_mine else cols.index_select(0, mine_t)
        blocks_in = [out[int(starts[p]):int(starts[p + 1])] for p in range(self.world)]
        through_rccl = self.always_collective          # the rank's own block as a send-to-self (tests the P2P path on one GPU)
        if not through_rccl:
            blocks_in[self.rank].copy_(rows_local[:, lo:hi] if all_of_mine else block_for(self.rank))
        if self.world > 1 or through_rccl:
            blocks_out = [block_for(p) if (p != self.rank or through_rccl) else rows_local[:0] for p in range(self.world)]
            self._exchange(name, blocks_out, blocks_in)
        plain = np.array_equal(row_index, np.arange(len(wanted_rows)))
        return out, (None if plain else row_index)

    def _maybe_gather(self, local_vec, gather, total=None):
        """All-gather of the column-sharded D-vector.  `total` (the full D) lets every rank compute every slice length;
        without it the lengths are exchanged first."""
        if not gather or not self._collective():
            return local_vec
        import torch
        if total is not None:
            lengths = [b - a for a, b in self.column_slices(int(total))]
            assert lengths[self.rank] == local_vec.shape[0]
        else:
            sizes = torch.tensor([local_vec.shape[0]], device=local_vec.device, dtype=torch.int64)
            all_sizes = [torch.zeros_like(sizes) for _ in range(self.world)]
            self.dist.all_gather(all_sizes, sizes, group=self.group)
            lengths = [int(s.item()) for s in all_sizes]
        width = max(lengths)
        padded = local_vec
        if local_vec.shape[0] != width:
            padded = torch.zeros(width, dtype=local_vec.dtype, device=local_vec.device)
            padded[:local_vec.shape[0]] = local_vec
        flat = torch.empty(width * self.world, dtype=local_vec.dtype, device=local_vec.device)
        self._timed('allgather_output', (self.world - 1) * width * local_vec.element_size(), padded,
                    lambda: self.dist.all_gather_into_tensor(flat, padded.contiguous(), group=self.group))
        if all(n == width for n in lengths):
            return flat
        return torch.cat([flat[p * width:p * width + n] for p, n in enumerate(lengths)])
