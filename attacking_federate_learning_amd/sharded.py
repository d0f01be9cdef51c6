"""Multi-GPU form of the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Layouts
  client-sharded  rank r holds rows [r*N/W, (r+1)*N/W) of G with all D columns -- where the clients'
                  gradients land (north_star).
  column-sharded  every rank holds all N rows for its own slice of the D columns.

The aggregation itself runs column-sharded, because that is the layout in which the path needs the least
exchange ("cheaper equivalent"):
  * trimmed_mean, no_defense and the attack statistics are independent per column: no collective at all;
  * the distance matrix is a sum over columns: each rank computes the fp64 Gram of its slice (MFMA kernel),
    ONE all-reduce of N x N doubles combines them (N = 10^4: 800 MB, vs 875 GB for all-gathering G), and
    every rank then runs the identical, deterministic selection on its own copy -- theta dependent argmins
    never cross the fabric;
  * the aggregate comes out column-sharded (optionally all-gathered: D*4 bytes).
`reshard_clients_to_columns` converts the client-sharded input with one personalised exchange (each rank
sends 1/W of its rows' columns to every peer: point-to-point over the xGMI full mesh, all links busy).

The all-reduce result is bitwise identical on every rank (each element is reduced in one fixed order and then
broadcast), so all ranks select the same clients; identical rows still tie exactly because the per-rank
partial Grams of identical rows are identical.

`LocalKernels` is the seam between this orchestration and the per-GPU kernels.  The product uses
`HipKernels` (libbyzagg on the rank's GPU).  The CPU test-suite drives the same orchestration over gloo with
a numpy stand-in defined in tests/ -- there is no CPU implementation in this package.
"""
import numpy as np


class HipKernels:
    """Per-rank kernels on the rank's own MI355X, operating on torch CUDA tensors."""

    def __init__(self, engine=None):
        from .engine import get_engine
        self.engine = engine or get_engine()

    def gram(self, g_local):
        return self.engine.gram(g_local)                       # (N, N) float64 CUDA tensor

    def distances_from_gram(self, gram):
        return self.engine.distances_from_gram(gram, gram.shape[0])

    def krum_select(self, dist, users_count, corrupted_count):
        return self.engine.krum_select(dist, users_count, corrupted_count)

    def bulyan_select(self, dist, users_count, corrupted_count):
        return self.engine.bulyan_select(dist, users_count, corrupted_count)

    def trimmed_mean(self, g_local, corrupted_count, row_index=None):
        return self.engine.trimmed_mean(g_local, g_local.shape[0], corrupted_count, row_index=row_index)

    def no_defense(self, g_local):
        return self.engine.no_defense(g_local)

    def drift(self, rows_local, num_std, write_back=False):
        return self.engine.drift_attack(rows_local, num_std, write_back=write_back)

    def row(self, g_local, index):
        return g_local[index].clone()


class ShardedAggregator:
    """defences.py / malicious.py over a column-sharded gradient matrix."""

    def __init__(self, kernels, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.kernels = kernels
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # BYZ_FORCE_COLLECTIVES=1 issues the collectives even at world size 1 (exercises the RCCL path on one GPU)
        import os
        self.always_collective = dist.is_initialized() and os.environ.get('BYZ_FORCE_COLLECTIVES') == '1'

    # ---- the one exchange step of the path -------------------------------------------------------
    def global_distances(self, g_local):
        gram = self.kernels.gram(g_local)
        if self.world > 1 or self.always_collective:
            self.dist.all_reduce(gram, op=self.dist.ReduceOp.SUM, group=self.group)
        return self.kernels.distances_from_gram(gram)

    # ---- defences.py -------------------------------------------------------------------------------
    def no_defense(self, g_local, users_count=None, corrupted_count=None, gather=False):
        return self._maybe_gather(self.kernels.no_defense(g_local), gather)

    def trimmed_mean(self, g_local, users_count, corrupted_count, gather=False):
        return self._maybe_gather(self.kernels.trimmed_mean(g_local, corrupted_count), gather)

    def krum(self, g_local, users_count, corrupted_count, return_index=False, gather=False):
        if not return_index:
            assert users_count >= 2 * corrupted_count + 1, (
                'users_count>=2*corrupted_count + 3', users_count, corrupted_count)
        dist_m = self.global_distances(g_local)
        index = self.kernels.krum_select(dist_m, users_count, corrupted_count)
        if return_index:
            return index
        return self._maybe_gather(self.kernels.row(g_local, index), gather)

    def bulyan(self, g_local, users_count, corrupted_count, gather=False, return_selection=False):
        assert users_count >= 4 * corrupted_count + 3
        dist_m = self.global_distances(g_local)
        selection = np.asarray(self.kernels.bulyan_select(dist_m, users_count, corrupted_count), dtype=np.int32)
        out = self.kernels.trimmed_mean(g_local, 2 * corrupted_count, row_index=selection)
        out = self._maybe_gather(out, gather)
        return (out, selection) if return_selection else out

    # ---- malicious.py ------------------------------------------------------------------------------
    def drift_attack(self, g_local, n_malicious, num_std, write_back=True, gather=False):
        """Rows 0..m-1 are the malicious clients (reference main.py:28); per column, no exchange."""
        drift, mean, std = self.kernels.drift(g_local[:n_malicious], num_std, write_back=write_back)
        return self._maybe_gather(drift, gather), mean, std

    # ---- layout conversion -------------------------------------------------------------------------
    def column_slices(self, n_cols):
        """Column range owned by every rank: contiguous, sizes differ by at most one."""
        base, extra = divmod(n_cols, self.world)
        bounds, start = [], 0
        for r in range(self.world):
            stop = start + base + (1 if r < extra else 0)
            bounds.append((start, stop))
            start = stop
        return bounds

    def reshard_clients_to_columns(self, rows_local, rows_per_rank):
        """client-sharded (rows_local: my rows x D) -> column-sharded (all rows x my columns).

        `rows_per_rank[r]` is the number of clients rank r holds.  One personalised exchange: point-to-point
        sends of (my rows) x (peer's columns); the received blocks are stacked in rank order.
        """
        import torch
        n_cols = rows_local.shape[1]
        bounds = self.column_slices(n_cols)
        lo, hi = bounds[self.rank]
        if self.world == 1:
            return rows_local[:, lo:hi].contiguous()
        out = torch.empty((int(sum(rows_per_rank)), hi - lo), dtype=rows_local.dtype, device=rows_local.device)
        offsets = np.concatenate([[0], np.cumsum(rows_per_rank)]).astype(int)
        ops, keep = [], []
        for peer in range(self.world):
            dst = out[offsets[peer]:offsets[peer + 1]]
            if peer == self.rank:
                dst.copy_(rows_local[:, lo:hi])
                continue
            plo, phi = bounds[peer]
            send = rows_local[:, plo:phi].contiguous()
            keep.append(send)
            global_peer = self.dist.get_global_rank(self.group, peer) if self.group is not None else peer
            ops.append(self.dist.P2POp(self.dist.isend, send, global_peer, group=self.group))
            ops.append(self.dist.P2POp(self.dist.irecv, dst, global_peer, group=self.group))
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()
        return out

    def _maybe_gather(self, local_vec, gather):
        if not gather or (self.world == 1 and not self.always_collective):
            return local_vec
        import torch
        sizes = torch.tensor([local_vec.shape[0]], device=local_vec.device, dtype=torch.int64)
        all_sizes = [torch.zeros_like(sizes) for _ in range(self.world)]
        self.dist.all_gather(all_sizes, sizes, group=self.group)
        lengths = [int(s.item()) for s in all_sizes]
        width = max(lengths)
        padded = torch.zeros(width, dtype=local_vec.dtype, device=local_vec.device)
        padded[:local_vec.shape[0]] = local_vec
        parts = [torch.empty_like(padded) for _ in range(self.world)]
        self.dist.all_gather(parts, padded, group=self.group)
        return torch.cat([p[:n] for p, n in zip(parts, lengths)])
