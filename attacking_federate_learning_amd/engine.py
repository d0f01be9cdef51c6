"""Host-side driver of libbyzagg: one `Engine` per GPU.

The engine accepts either
  * host matrices (C-contiguous np.float32, exactly what reference server.py:34-35 allocates): staged to
    the device by the library, results come back as numpy -- this is the drop-in path; or
  * device-resident matrices (torch CUDA tensors, or `DeviceBuffer`s allocated through the library):
    nothing crosses PCIe, work is queued on the caller's current stream -- this is the path the large
    configurations and bench.py use.
There is no CPU implementation behind it: every method ends in a HIP kernel or raises.
"""
import ctypes
import itertools
import os

import numpy as np

from . import _native

NAME_TO_ID = {'NoDefense': 0, 'Krum': 1, 'TrimmedMean': 2, 'Bulyan': 3}


class EngineError(RuntimeError):
    pass


def _check(rc):
    if rc == _native.OK:
        return
    msg = _native.last_error()
    if rc == _native.E_PRECONDITION:
        raise AssertionError(msg)            # the reference asserts (defences.py:25, 56)
    if rc == _native.E_NO_WINNER:
        raise KeyError(-1)                   # the reference pops key -1 from its dict (defences.py:65)
    if rc == _native.E_INVALID:
        raise ValueError(msg)
    if rc == _native.E_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise EngineError('libbyzagg error %d: %s' % (rc, msg))


def _is_torch(x):
    return type(x).__module__.split('.')[0] == 'torch'


def _vp(x):
    return ctypes.c_void_p(int(x)) if x else None


class DeviceBuffer:
    """Raw device memory owned through the C ABI (lets a host without torch keep data on the GPU)."""

    def __init__(self, engine, shape, dtype):
        self.engine = engine
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        ptr = ctypes.c_void_p()
        _check(engine.lib.byz_malloc(engine.ctx, max(self.nbytes, 4), ctypes.byref(ptr)))
        self.ptr = ptr.value

    def upload(self, array):
        array = np.ascontiguousarray(array, dtype=self.dtype)
        assert array.nbytes == self.nbytes
        _check(self.engine.lib.byz_upload(self.engine.ctx, _vp(self.ptr), array.ctypes.data_as(ctypes.c_void_p),
                                          self.nbytes, None))
        _check(self.engine.lib.byz_stream_sync(self.engine.ctx, None))
        return self

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _check(self.engine.lib.byz_download(self.engine.ctx, out.ctypes.data_as(ctypes.c_void_p), _vp(self.ptr),
                                            self.nbytes, None))
        return out

    def free(self):
        if self.ptr and self.engine.ctx:
            self.engine.lib.byz_free(self.engine.ctx, _vp(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Distances:
    """What `_krum_create_distances` returns here: the N x N fp32 distance matrix, resident on the GPU.

    The reference returns a dict of dicts (defences.py:16-21); `to_dict()` rebuilds that form, with the same
    key order (1, 0, 2, ...), for code that wants to look inside.
    """

    def __init__(self, buffer, n):
        self.buffer, self.n = buffer, n

    @property
    def ptr(self):
        return self.buffer.ptr

    def numpy(self):
        return self.buffer.numpy()

    def to_dict(self):
        from collections import defaultdict
        dense = self.numpy()
        out = defaultdict(dict)
        for i in range(self.n):
            for j in range(i):
                out[i][j] = out[j][i] = dense[i, j]
        return out


class _Matrix:
    """Uniform view of a caller's matrix: device pointer, shape, leading dimension, stream."""

    def __init__(self, ptr, rows, cols, ld, stream, keepalive, torch_like=None):
        self.ptr, self.rows, self.cols, self.ld, self.stream = ptr, rows, cols, ld, stream
        self.keepalive, self.torch_like = keepalive, torch_like


class Engine:
    def __init__(self, device=None):
        self.lib = _native.load()
        if device is None:
            device = int(os.environ.get('BYZ_DEVICE', os.environ.get('LOCAL_RANK', '0')))
        ctx = ctypes.c_void_p()
        _check(self.lib.byz_ctx_create(int(device), ctypes.byref(ctx)))
        self.ctx = ctx
        self.device = int(device)
        self._assemble_key = None     # what the context's device-side pointer table holds (assemble_rows)

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.byz_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory helpers -----------------------------------------------------------------------
    def empty(self, shape, dtype=np.float32):
        return DeviceBuffer(self, shape, dtype)

    def to_device(self, array):
        array = np.ascontiguousarray(array)
        return DeviceBuffer(self, array.shape, array.dtype).upload(array)

    def reserve(self, n_rows, n_cols):
        _check(self.lib.byz_ctx_reserve(self.ctx, int(n_rows), int(n_cols)))

    def synchronize(self, stream=None):
        _check(self.lib.byz_stream_sync(self.ctx, _vp(stream)))

    def check(self, stream=None):
        """Synchronise and raise if a kernel of an asynchronous call flagged a failure on the device."""
        _check(self.lib.byz_ctx_check(self.ctx, _vp(stream)))

    def _device_matrix(self, g):
        """torch CUDA tensor / DeviceBuffer -> _Matrix; None for host arrays."""
        if isinstance(g, DeviceBuffer):
            assert g.dtype == np.float32 and len(g.shape) == 2
            return _Matrix(g.ptr, g.shape[0], g.shape[1], g.shape[1], None, g)
        if _is_torch(g):
            import torch
            if not g.is_cuda:
                return None
            if g.dtype != torch.float32 or g.dim() != 2 or g.stride(1) != 1:
                raise ValueError('device matrices must be 2-D float32 with unit column stride')
            if g.device.index != self.device:
                raise ValueError('the matrix lives on %s, this engine drives cuda:%d' % (g.device, self.device))
            stream = torch.cuda.current_stream(g.device).cuda_stream
            return _Matrix(g.data_ptr(), g.shape[0], g.shape[1], g.stride(0), stream, g, torch_like=g)
        return None

    @staticmethod
    def _host_matrix(g):
        if _is_torch(g):
            g = g.detach().cpu().numpy()
        g = np.asarray(g)
        if g.ndim != 2:
            raise ValueError('expected a 2-D gradient matrix, got shape %r' % (g.shape,))
        return np.ascontiguousarray(g, dtype=np.float32)

    def _out_like(self, m, n, dtype=np.float32):
        """Output vector for a device-resident input: torch tensor for torch input, DeviceBuffer otherwise."""
        if m.torch_like is not None:
            import torch
            tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32}[np.dtype(dtype)]
            t = torch.empty(int(n), dtype=tdt, device=m.torch_like.device)
            return t, t.data_ptr()
        b = DeviceBuffer(self, (int(n),), dtype)
        return b, b.ptr

    # ---- defences.py -------------------------------------------------------------------------
    def defend_host(self, name, g, users_count, corrupted_count, check_assert=True, want_aux=False):
        """One call for a host matrix: upload, aggregate, download (the numpy drop-in path)."""
        g = self._host_matrix(g)
        n, d = g.shape
        out = np.empty(d, dtype=np.float32)
        theta = max(int(users_count) - 2 * int(corrupted_count), 1)
        aux = np.full(max(theta, 1), -1, dtype=np.int32) if want_aux else None
        _check(self.lib.byz_defend_host(self.ctx, NAME_TO_ID[name], g.ctypes.data_as(ctypes.c_void_p), n, d,
                                        int(users_count), int(corrupted_count), int(bool(check_assert)),
                                        out.ctypes.data_as(ctypes.c_void_p),
                                        aux.ctypes.data_as(ctypes.c_void_p) if want_aux else None))
        return (out, aux) if want_aux else out

    def no_defense(self, g, users_count=None, corrupted_count=None):
        m = self._device_matrix(g)
        if m is None:
            return self.defend_host('NoDefense', g, 0, 0)
        out, ptr = self._out_like(m, m.cols)
        _check(self.lib.byz_no_defense_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, _vp(ptr), _vp(m.stream)))
        return out

    def pairwise_distances(self, g):
        m = self._device_matrix(g)
        stage = None
        if m is None:
            stage = self.to_device(self._host_matrix(g))
            m = self._device_matrix(stage)
        dist = DeviceBuffer(self, (m.rows, m.rows), np.float32)
        _check(self.lib.byz_pairwise_distances_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, _vp(dist.ptr),
                                                   _vp(m.stream)))
        self.check(m.stream)
        return Distances(dist, m.rows)

    def gram(self, g):
        """fp64 Gram matrix of a device-resident slice (the D-sharded multi-GPU path all-reduces these)."""
        m = self._device_matrix(g)
        if m is None:
            raise ValueError('gram() takes a device-resident matrix')
        if m.torch_like is not None:
            import torch
            out = torch.empty((m.rows, m.rows), dtype=torch.float64, device=m.torch_like.device)
            ptr = out.data_ptr()
        else:
            out = DeviceBuffer(self, (m.rows, m.rows), np.float64)
            ptr = out.ptr
        _check(self.lib.byz_gram_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, _vp(ptr), _vp(m.stream)))
        return out

    def gram_share_add(self, panel, row_index, share_count, share_index, gram):
        """`gram` (the caller's N x N fp64 device buffer, zeroed before the first panel) += this rank's share of the Gram
        tiles of `panel`: the per-panel N x N addition pass of the clients layout folded into the reduction kernel."""
        m = self._device_matrix(panel)
        if m is None:
            raise ValueError('gram_share_add() takes a device-resident matrix')
        n_rows, idx_ptr, keep = m.rows, None, None
        if row_index is not None:
            idx_ptr, n_rows, keep = self._row_index(row_index, m, validate=False)
        if tuple(gram.shape) != (n_rows, n_rows):
            raise ValueError('the accumulator must be %d x %d' % (n_rows, n_rows))
        # the kernel read-modify-writes N * N doubles at this address: anything else than a contiguous fp64 buffer on the
        # panel's device would corrupt memory silently
        if _is_torch(gram):
            import torch
            if gram.dtype != torch.float64 or not gram.is_contiguous() or not gram.is_cuda:
                raise ValueError('the accumulator must be a contiguous float64 CUDA tensor')
            if m.torch_like is not None and gram.device != m.torch_like.device:
                raise ValueError('the accumulator lives on %s, the panel on %s' % (gram.device, m.torch_like.device))
            ptr = gram.data_ptr()
        elif isinstance(gram, DeviceBuffer):
            if gram.dtype != np.float64:
                raise ValueError('the accumulator must be float64')
            ptr = gram.ptr
        else:
            raise ValueError('the accumulator must be a torch CUDA tensor or a DeviceBuffer')
        _check(self.lib.byz_gram_share_add_dev(self.ctx, _vp(m.ptr), int(n_rows), m.cols, m.ld, _vp(idx_ptr),
                                               int(share_count), int(share_index), _vp(ptr), _vp(m.stream)))
        if keep is not None and not _is_torch(keep):
            self.synchronize(m.stream)
        return gram

    def gram_share(self, panel, row_index, share_count, share_index):
        """This rank's share of the Gram tiles of `panel` (a device matrix every rank holds, e.g. an all-gathered column
        panel): zeros outside the share, so the ranks' outputs SUM to the Gram.  `row_index` (int32, on the device, or
        None): logical row r is panel[row_index[r]] -- skips the padding rows of an unevenly sharded gather."""
        m = self._device_matrix(panel)
        if m is None:
            raise ValueError('gram_share() takes a device-resident matrix')
        n_rows, idx_ptr, keep = m.rows, None, None
        if row_index is not None:
            idx_ptr, n_rows, keep = self._row_index(row_index, m, validate=False)
        if m.torch_like is not None:
            import torch
            out = torch.empty((n_rows, n_rows), dtype=torch.float64, device=m.torch_like.device)
            ptr = out.data_ptr()
        else:
            out = DeviceBuffer(self, (n_rows, n_rows), np.float64)
            ptr = out.ptr
        _check(self.lib.byz_gram_share_dev(self.ctx, _vp(m.ptr), int(n_rows), m.cols, m.ld, _vp(idx_ptr),
                                           int(share_count), int(share_index), _vp(ptr), _vp(m.stream)))
        if keep is not None and not _is_torch(keep):
            self.synchronize(m.stream)
        return out

    def near_pairs_count(self, stream=None):
        """Pairs the last distance kernel could not resolve through the Gram identity (synchronises)."""
        count = ctypes.c_int64(0)
        _check(self.lib.byz_near_pairs_count(self.ctx, ctypes.byref(count), _vp(stream)))
        return int(count.value)

    def near_pairs_sqdist(self, g, count, row_index=None):
        """Per listed pair, the sum over g's columns of the squared fp32 difference (fp64): this rank's part."""
        m = self._device_matrix(g)
        if m is None:
            raise ValueError('near_pairs_sqdist() takes a device-resident matrix')
        idx_ptr, keep = None, None
        if row_index is not None:
            idx_ptr, _, keep = self._row_index(row_index, m, validate=False)
        if m.torch_like is not None:
            import torch
            sq = torch.empty(int(count), dtype=torch.float64, device=m.torch_like.device)
            sq_ptr = sq.data_ptr()
        else:
            sq = DeviceBuffer(self, (int(count),), np.float64)
            sq_ptr = sq.ptr
        _check(self.lib.byz_near_pairs_sqdist_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, _vp(idx_ptr), _vp(sq_ptr),
                                                  _vp(m.stream)))
        if keep is not None and not _is_torch(keep):
            self.synchronize(m.stream)
        return sq

    def near_pairs_apply(self, sq, distances, stream=None):
        """distances[i, j] = sqrt(sq[p]) for every listed pair, then identical rows are made to tie exactly again."""
        ptr = sq.data_ptr() if _is_torch(sq) else sq.ptr
        if _is_torch(sq):
            import torch
            stream = torch.cuda.current_stream(sq.device).cuda_stream
        _check(self.lib.byz_near_pairs_apply_dev(self.ctx, _vp(ptr), int(distances.n), _vp(distances.ptr), _vp(stream)))
        self.check(stream)

    def distances_from_gram(self, gram, n, stream=None, local_columns=None, all_reduce=None):
        """Distances from an (all-reduced) fp64 Gram.  The Gram identity cannot resolve nearly coincident rows, so the
        kernel lists those pairs; with `local_columns` (this rank's column slice of G) they are re-evaluated on the
        difference itself: per-rank sums of squared differences, `all_reduce(tensor)` over the ranks when given, then
        applied -- the same result as the single-GPU path, which does all of it inside byz_pairwise_distances_dev.
        Without `local_columns` the list is left for the caller (near_pairs_count / _sqdist / _apply)."""
        ptr = gram.data_ptr() if _is_torch(gram) else gram.ptr
        if _is_torch(gram):
            import torch
            stream = torch.cuda.current_stream(gram.device).cuda_stream
        dist = Distances(DeviceBuffer(self, (n, n), np.float32), n)
        _check(self.lib.byz_distances_from_gram_dev(self.ctx, _vp(ptr), int(n), _vp(dist.ptr), _vp(stream)))
        if local_columns is None:
            self.check(stream)
            return dist
        count = self.near_pairs_count(stream)
        if count:
            sq = self.near_pairs_sqdist(local_columns, count)
            if all_reduce is not None:
                all_reduce(sq)
            self.near_pairs_apply(sq, dist, stream)
        return dist

    def _as_distances(self, distances):
        """-> (Distances, keys): `keys[c]` is the reference's row index of row c of the dense matrix, or None when
        they coincide.

        A dict is what the reference's own loops pass around (defences.py:16-21, and :61-68 where Bulyan pops the
        winner's row and column before the next `krum(..., distances, True)`): any subset of rows may be present,
        in dict order.  The kernels visit rows in the order 1, 0, 2, 3, ..., which IS the dict order of a complete
        dict; for a dict with rows removed the dense matrix lists the keys in dict order with the first two
        swapped, so that the same kernel order walks them as the reference's `for user in distances.keys()` does."""
        if isinstance(distances, Distances):
            return distances, None
        if isinstance(distances, dict):
            keys = list(distances.keys())
            m = len(keys)
            if m >= 2:
                keys[0], keys[1] = keys[1], keys[0]
            position = {k: c for c, k in enumerate(keys)}
            dense = np.full((m, m), np.inf, dtype=np.float32)
            for k, row in distances.items():
                if len(row) != m - 1:
                    raise ValueError('distance dict is not square: row %r has %d entries, %d rows present'
                                     % (k, len(row), m))
                c = position[k]
                for j, v in row.items():
                    dense[c, position[j]] = v
            plain = keys == list(range(m))
            return Distances(self.to_device(dense), m), (None if plain else keys)
        dense = np.ascontiguousarray(distances, dtype=np.float32)
        if dense.ndim != 2 or dense.shape[0] != dense.shape[1]:
            raise ValueError('distances must be a square matrix')
        return Distances(self.to_device(dense), dense.shape[0]), None

    def krum_select(self, distances, users_count, corrupted_count):
        """The selection loop of defences.py:27-37 on a distance matrix; returns the index (or -1)."""
        d, keys = self._as_distances(distances)
        idx = ctypes.c_int32(-2)
        _check(self.lib.byz_krum_select_dev(self.ctx, _vp(d.ptr), d.n, int(users_count), int(corrupted_count),
                                            ctypes.byref(idx), None, None))
        idx = int(idx.value)
        return idx if (keys is None or idx < 0) else keys[idx]

    def krum(self, g, users_count, corrupted_count, distances=None, return_index=False, debug=False, *, check=True):
        """defences.krum (defences.py:23-42).  Device-resident `g` without `return_index`: the winning row is copied on
        the device and nothing needs to cross to the host, but a kernel can only FLAG a failure there (a Gram chunk that
        never got its ticket, helpers that lost their worker: the sticky status word) -- so by default the call ends with
        `self.check(stream)` (one synchronisation) and raises instead of returning a row picked from invalid distances.
        `check=False` (keyword only: the positional slot behind `return_index` is the reference's `debug`, accepted and
        ignored) keeps the call asynchronous; the caller then owes an `engine.check()` before it uses the result."""
        if not return_index:
            # defences.py:24-25 (the message says +3, the check is +1)
            assert users_count >= 2 * corrupted_count + 1, (
                'users_count>=2*corrupted_count + 3', users_count, corrupted_count)
        if distances is not None:
            idx = self.krum_select(distances, users_count, corrupted_count)
            if return_index:
                return idx
            return self._row(g, idx)
        m = self._device_matrix(g)
        if m is None:
            out, aux = self.defend_host('Krum', g, users_count, corrupted_count, check_assert=False, want_aux=True)
            if return_index:
                return int(aux[0])
            if isinstance(g, np.ndarray) and g.ndim == 2:
                # the reference returns users_grads[index] -- a VIEW of the caller's matrix (defences.py:42), -1 = numpy's last
                # row when no score beat 1e20 -- and since round 6 so does the host path (missing 6)
                return g[int(aux[0])]
            return out
        idx = ctypes.c_int32(-2)
        out, ptr = (None, None) if return_index else self._out_like(m, m.cols)
        # the winning row is copied on the device; the index crosses to the host (one sync) only when asked for
        _check(self.lib.byz_krum_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, int(users_count),
                                     int(corrupted_count), 0, _vp(ptr), ctypes.byref(idx) if return_index else None,
                                     _vp(m.stream)))
        if return_index:
            return int(idx.value)      # (the library read the status word together with the index)
        if check:
            self.check(m.stream)
        return out

    def _row(self, g, idx):
        """Row `idx` of the caller's matrix, in the caller's own container type (negative idx as numpy)."""
        if isinstance(g, DeviceBuffer):
            return g.numpy()[idx]
        return g[idx]

    def _row_index(self, row_index, m, validate):
        """row_index -> (device pointer, count, keepalive).  The kernel reads int32 indices straight from device
        memory: anything else is converted here, and (unless the caller vouches for it) checked against the
        matrix height, because an out-of-range index is an out-of-bounds read on the GPU."""
        if isinstance(row_index, DeviceBuffer):
            if row_index.dtype != np.int32:
                raise ValueError('a DeviceBuffer row_index must hold int32')
            return row_index.ptr, int(np.prod(row_index.shape)), row_index
        if _is_torch(row_index):
            import torch
            if not row_index.is_cuda or (m.torch_like is not None and row_index.device != m.torch_like.device) \
                    or row_index.device.index != self.device:
                raise ValueError('row_index must live on the GPU of the matrix (cuda:%d), got %s'
                                 % (self.device, row_index.device))
            if row_index.dtype not in (torch.int32, torch.int64, torch.int16, torch.uint8):
                raise ValueError('row_index must hold integers, got %s' % row_index.dtype)
            idx = row_index.reshape(-1).to(torch.int32).contiguous()
            if validate and idx.numel():
                lo, hi = int(idx.min().item()), int(idx.max().item())
                if lo < 0 or hi >= m.rows:
                    raise ValueError('row_index out of range: [%d, %d] for %d rows' % (lo, hi, m.rows))
            return idx.data_ptr(), idx.numel(), idx
        host = np.asarray(row_index)
        if host.dtype.kind not in 'iu':
            raise ValueError('row_index must hold integers, got %s' % host.dtype)
        host = host.reshape(-1)
        if host.size and (host.min() < 0 or host.max() >= m.rows):
            raise ValueError('row_index out of range: [%d, %d] for %d rows' % (host.min(), host.max(), m.rows))
        keep = self.to_device(host.astype(np.int32))
        return keep.ptr, keep.shape[0], keep

    def trimmed_mean(self, g, users_count=None, corrupted_count=0, row_index=None, validate_index=True):
        m = self._device_matrix(g)
        if m is None:
            if row_index is not None:
                g = self._host_matrix(g)[np.asarray(row_index)]
            return self.defend_host('TrimmedMean', g, 0, corrupted_count)
        n_rows, idx_ptr, keep = m.rows, None, None
        if row_index is not None:
            idx_ptr, n_rows, keep = self._row_index(row_index, m, validate_index)
            if n_rows == 0:
                raise ValueError('row_index selects no rows')
        out, ptr = self._out_like(m, m.cols)
        _check(self.lib.byz_trimmed_mean_dev(self.ctx, _vp(m.ptr), int(n_rows), m.cols, m.ld, _vp(idx_ptr),
                                             int(corrupted_count), _vp(ptr), _vp(m.stream)))
        if keep is not None and not _is_torch(keep):
            self.synchronize(m.stream)  # the temporary index buffer must outlive the kernel
        return out

    def trimmed_mean_redone(self, stream=None):
        """16-column tiles of the last trimmed mean that the ring selection handed to the general kernel."""
        tiles = ctypes.c_int64(0)
        _check(self.lib.byz_trimmed_mean_redone(self.ctx, ctypes.byref(tiles), _vp(stream)))
        return int(tiles.value)

    def bulyan_select(self, distances, users_count, corrupted_count, on_device=False):
        """The pick-and-remove loop of defences.py:59-68: theta indices in selection order.  `on_device=True` leaves
        them in a DeviceBuffer (int32) for `trimmed_mean(row_index=...)` -- no host round trip between the stages."""
        d, keys = self._as_distances(distances)
        theta = int(users_count) - 2 * int(corrupted_count)
        sel = DeviceBuffer(self, (max(theta, 1),), np.int32)
        _check(self.lib.byz_bulyan_select_dev(self.ctx, _vp(d.ptr), d.n, int(users_count), int(corrupted_count),
                                              _vp(sel.ptr), None))
        if on_device and keys is None:
            sel.shape = (theta,)
            sel.nbytes = theta * 4
            return sel
        picked = sel.numpy()[:theta]
        return picked if keys is None else np.asarray([keys[i] for i in picked], dtype=np.int32)

    def krum_bulyan_select(self, distances, users_count, corrupted_count, on_device=False):
        """(Krum index, Bulyan selection) from ONE distance matrix with one sort of its rows -- what a round that runs both
        defences on the same distances needs (BASELINE configs[4]); the same values `krum_select` and `bulyan_select` return."""
        d, keys = self._as_distances(distances)
        theta = int(users_count) - 2 * int(corrupted_count)
        sel = DeviceBuffer(self, (max(theta, 1),), np.int32)
        idx = ctypes.c_int32(-2)
        _check(self.lib.byz_krum_bulyan_select_dev(self.ctx, _vp(d.ptr), d.n, int(users_count), int(corrupted_count),
                                                   ctypes.byref(idx), _vp(sel.ptr), None))
        index = int(idx.value)
        index = index if (keys is None or index < 0) else keys[index]
        if on_device and keys is None:
            sel.shape = (theta,)
            sel.nbytes = theta * 4
            return index, sel
        picked = sel.numpy()[:theta]
        return index, (picked if keys is None else np.asarray([keys[i] for i in picked], dtype=np.int32))

    def bulyan_rescored(self):
        """Rows the last Bulyan loop re-scored with the reference's sequential fp32 sums (0: every pick was clear)."""
        rows = ctypes.c_int64(0)
        _check(self.lib.byz_bulyan_rescored(self.ctx, ctypes.byref(rows)))
        return int(rows.value)

    def bulyan(self, g, users_count, corrupted_count, return_selection=False):
        assert users_count >= 4 * corrupted_count + 3  # defences.py:56
        m = self._device_matrix(g)
        if m is None:
            if return_selection:
                out, aux = self.defend_host('Bulyan', g, users_count, corrupted_count, want_aux=True)
                return out, aux[:int(users_count) - 2 * int(corrupted_count)]
            return self.defend_host('Bulyan', g, users_count, corrupted_count)
        theta = int(users_count) - 2 * int(corrupted_count)
        out, ptr = self._out_like(m, m.cols)
        sel, sel_ptr = self._out_like(m, max(theta, 1), np.int32) if return_selection else (None, None)
        _check(self.lib.byz_bulyan_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, int(users_count),
                                       int(corrupted_count), _vp(ptr), _vp(sel_ptr), _vp(m.stream)))
        return (out, sel) if return_selection else out

    # ---- malicious.py ------------------------------------------------------------------------
    def drift_attack(self, rows, num_std, write_back=False):
        """(drift, mean, std) over the rows; device inputs may be overwritten in place (write_back)."""
        m = self._device_matrix(rows)
        if m is None:
            g = self._host_matrix(rows)
            n, d = g.shape
            drift, mean, std = (np.empty(d, dtype=np.float32) for _ in range(3))
            _check(self.lib.byz_drift_attack_host(self.ctx, g.ctypes.data_as(ctypes.c_void_p), n, d, float(num_std),
                                                  drift.ctypes.data_as(ctypes.c_void_p),
                                                  mean.ctypes.data_as(ctypes.c_void_p),
                                                  std.ctypes.data_as(ctypes.c_void_p)))
            return drift, mean, std
        outs = [self._out_like(m, m.cols) for _ in range(3)]
        _check(self.lib.byz_drift_attack_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, float(num_std),
                                             _vp(outs[0][1]), _vp(outs[1][1]), _vp(outs[2][1]),
                                             int(bool(write_back)), _vp(m.stream)))
        return outs[0][0], outs[1][0], outs[2][0]

    def column_chain(self, rows, carry=None, mean=None):
        """One link of the attack's statistics over rows that several owners hold (sharded.py, clients layout): the running
        column sums after THESE rows, continuing `carry` (the previous owner's result; None: the chain starts here) -- of the
        values, or with `mean` of their squared fp32 deviations.  Device tensors in, a device vector out."""
        m = self._device_matrix(rows)
        if m is None or m.torch_like is None:
            raise ValueError('column_chain() takes a torch CUDA matrix')
        import torch
        for v in (carry, mean):
            if v is not None and (not _is_torch(v) or v.dtype != torch.float32 or not v.is_contiguous() or v.numel() != m.cols
                                  or v.device != m.torch_like.device):
                raise ValueError('carry / mean must be contiguous float32 vectors of %d columns on the matrix\'s device' % m.cols)
        out = torch.empty(m.cols, dtype=torch.float32, device=m.torch_like.device)
        _check(self.lib.byz_column_chain_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld,
                                             _vp(carry.data_ptr() if carry is not None else None),
                                             _vp(mean.data_ptr() if mean is not None else None), _vp(out.data_ptr()), _vp(m.stream)))
        return out

    def column_finish(self, total_rows, num_std, sum=None, sumsq=None, mean=None):
        """The end of a chain.  sum given: returns mean = sum / total_rows.  sumsq (and mean) given: returns (std, drift) =
        (sqrt(sumsq / total_rows), mean - num_std * std)."""
        import torch
        if (sum is None) == (sumsq is None):
            raise ValueError('column_finish: give the sum (returns the mean) or the sum of squared deviations (returns std, drift)')
        if sum is None and mean is None:
            raise ValueError('column_finish: the squared deviations need the mean they were taken about')
        ref = sum if sum is not None else sumsq
        for name, t in (('sum', sum), ('sumsq', sumsq), ('mean', mean)):
            if t is None:
                continue
            if not (_is_torch(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError('column_finish: %s must be a contiguous float32 CUDA tensor' % name)
            if t.numel() != ref.numel() or t.device != ref.device:
                raise ValueError('column_finish: %s has %d values on %s, expected %d on %s' % (name, t.numel(), t.device, ref.numel(), ref.device))
        n = ref.numel()
        stream = torch.cuda.current_stream(ref.device).cuda_stream
        if sum is not None:
            mean = torch.empty_like(sum)
            _check(self.lib.byz_column_finish_dev(self.ctx, _vp(sum.data_ptr()), None, int(total_rows), float(num_std), n,
                                                  _vp(mean.data_ptr()), None, None, _vp(stream)))
            return mean
        std, drift = torch.empty_like(sumsq), torch.empty_like(sumsq)
        _check(self.lib.byz_column_finish_dev(self.ctx, None, _vp(sumsq.data_ptr()), int(total_rows), float(num_std), n,
                                              _vp(mean.data_ptr()), _vp(std.data_ptr()), _vp(drift.data_ptr()), _vp(stream)))
        return std, drift

    def drift_axpy_host(self, mean, std, num_std):
        """DriftAttack._attack_grads on host vectors (malicious.py:34-36), computed on the device."""
        mean_c = np.ascontiguousarray(mean, dtype=np.float32)
        a, b = self.to_device(mean_c), self.to_device(np.ascontiguousarray(std, dtype=np.float32))
        _check(self.lib.byz_drift_axpy_dev(self.ctx, _vp(a.ptr), _vp(b.ptr), mean_c.size, float(num_std), None))
        return a.numpy()

    # ---- server.py:89-90 ---------------------------------------------------------------------
    def server_update(self, weights, velocity, agg, momentum, learning_rate):
        """In-place fused momentum update on device-resident vectors (torch tensors or DeviceBuffers)."""
        def ptr_of(x):
            return x.data_ptr() if _is_torch(x) else x.ptr
        n = weights.numel() if _is_torch(weights) else int(np.prod(weights.shape))
        stream = None
        if _is_torch(weights):
            import torch
            stream = torch.cuda.current_stream(weights.device).cuda_stream
        _check(self.lib.byz_server_update_dev(self.ctx, _vp(ptr_of(weights)), _vp(ptr_of(velocity)), _vp(ptr_of(agg)),
                                              int(n), float(momentum), float(learning_rate), _vp(stream)))

    # ---- backdoor.py:52-65 (the hook's arithmetic; the training loop stays with the caller) ----
    def _vectors(self, *vectors):
        """Same-length fp32 vectors -> (device pointers, n, stream, keepalives, torch example or None).
        Host vectors (numpy) are uploaded; torch CUDA tensors and DeviceBuffers are used in place."""
        ptrs, keep, stream, example = [], [], None, None
        n = None
        for v in vectors:
            if isinstance(v, DeviceBuffer):
                assert v.dtype == np.float32
                size, ptr = int(np.prod(v.shape)), v.ptr
            elif _is_torch(v) and v.is_cuda:
                import torch
                if v.dtype != torch.float32 or not v.is_contiguous():
                    raise ValueError('device vectors must be contiguous float32')
                size, ptr, example = v.numel(), v.data_ptr(), v
                stream = torch.cuda.current_stream(v.device).cuda_stream
            else:
                host = v.detach().cpu().numpy() if _is_torch(v) else np.asarray(v)
                buf = self.to_device(np.ascontiguousarray(host, dtype=np.float32).ravel())
                keep.append(buf)
                size, ptr = int(np.prod(buf.shape)), buf.ptr
            if n is None:
                n = size
            elif size != n:
                raise ValueError('vector lengths differ: %d and %d' % (n, size))
            ptrs.append(ptr)
            keep.append(v)
        return ptrs, n, stream, keep, example

    def _vector_out(self, n, example):
        if example is not None:
            import torch
            t = torch.empty(int(n), dtype=torch.float32, device=example.device)
            return t, t.data_ptr()
        b = DeviceBuffer(self, (int(n),), np.float32)
        return b, b.ptr

    def backdoor_initial_params(self, original_params, grads_mean, learning_rate):
        """original_params - learning_rate * grads_mean (backdoor.py:54).  numpy in -> numpy out;
        device-resident in -> device-resident out."""
        (p, m), n, stream, keep, example = self._vectors(original_params, grads_mean)
        out, optr = self._vector_out(n, example)
        _check(self.lib.byz_backdoor_initial_params_dev(self.ctx, _vp(p), _vp(m), n, float(np.float32(learning_rate)),
                                                        _vp(optr), _vp(stream)))
        host = example is None and not any(isinstance(v, DeviceBuffer) for v in (original_params, grads_mean))
        return out.numpy() if host else out

    def backdoor_clip(self, grads_mean, grads_stdev, original_params, mal_net_params, learning_rate, num_std):
        """The gradient that leads to `mal_net_params`, clipped to mean +- num_std * std (backdoor.py:57-63)."""
        args = (grads_mean, grads_stdev, original_params, mal_net_params)
        (m, s, p, q), n, stream, keep, example = self._vectors(*args)
        out, optr = self._vector_out(n, example)
        _check(self.lib.byz_backdoor_clip_dev(self.ctx, _vp(m), _vp(s), _vp(p), _vp(q), n,
                                              float(np.float32(learning_rate)), float(np.float32(num_std)),
                                              _vp(optr), _vp(stream)))
        host = example is None and not any(isinstance(v, DeviceBuffer) for v in args)
        return out.numpy() if host else out

    # ---- user.py:92 + server.py:81-83 --------------------------------------------------------------
    def assemble_row(self, g, row, grads):
        """Row `row` of the device-resident matrix `g` := the client's gradient.

        `grads` is either the client's flat vector (numpy, what `usr.grads` is in the reference: one
        host-to-device copy) or a sequence of device tensors, one per model parameter in parameter order
        (torch CUDA tensors or DeviceBuffers): those are concatenated straight into the row by one kernel,
        and the gradient never visits the host."""
        m = self._device_matrix(g)
        if m is None:
            raise ValueError('assemble_row() fills a device-resident matrix')
        if isinstance(grads, (list, tuple)):
            ptrs, lens, keep = [], [], []
            for t in grads:
                if isinstance(t, DeviceBuffer):
                    assert t.dtype == np.float32
                    ptrs.append(t.ptr)
                    lens.append(int(np.prod(t.shape)))
                elif _is_torch(t) and t.is_cuda:
                    import torch
                    if t.dtype != torch.float32:
                        raise ValueError('gradient tensors must be float32')
                    t = t if t.is_contiguous() else t.contiguous()
                    ptrs.append(t.data_ptr())
                    lens.append(t.numel())
                else:
                    raise ValueError('a list of gradients must hold device tensors; pass host data as one flat vector')
                keep.append(t)
            sync_after = self._order_after_torch(m, keep)
            table = (ctypes.c_void_p * len(ptrs))(*ptrs)
            lengths = (ctypes.c_int64 * len(lens))(*lens)
            _check(self.lib.byz_assemble_row_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, int(row), len(ptrs),
                                                 table, lengths, _vp(m.stream)))
            if sync_after:
                self.synchronize(m.stream)
            return
        host = grads.detach().cpu().numpy() if _is_torch(grads) else np.asarray(grads)
        host = np.ascontiguousarray(host, dtype=np.float32).ravel()
        if host.size != m.cols:
            raise ValueError('the row holds %d values, the gradient %d' % (m.cols, host.size))
        _check(self.lib.byz_assemble_row_host(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, int(row),
                                              host.ctypes.data_as(ctypes.c_void_p), _vp(m.stream)))
        self.synchronize(m.stream)   # `host` may be a temporary

    def _order_after_torch(self, m, tensors):
        """The launch goes to m.stream; torch tensors (and the .contiguous() temporaries made of them) belong to torch's
        current stream.  Where the two differ -- a DeviceBuffer matrix filled from torch tensors under a side stream --
        the producers are waited for before the launch and the launch before the temporaries can be freed.
        Returns whether the caller must synchronise m.stream after the launch."""
        for t in tensors:
            if _is_torch(t):
                import torch
                current = torch.cuda.current_stream(t.device)
                if int(current.cuda_stream) != int(m.stream or 0):
                    current.synchronize()
                    return True
                return False
        return False

    def assemble_rows(self, g, first_row, clients_grads):
        """Rows first_row .. of the device-resident matrix := the clients' gradients, ONE launch for all of them
        (server.py:81-83's loop).  `clients_grads[c]` is client c's sequence of device tensors, one per model parameter in
        parameter order; every client has the same parameter sizes (one model).

        The table of the clients' tensor addresses is built and uploaded when an address changed since the last call; a round
        in which no tensor moved (the normal case: a model's .grad buffers keep their addresses) costs one pass over
        `data_ptr()` and the launch (the table of 400 tensors was rebuilt and uploaded every round, twelve
        times the kernel's own time)."""
        m = self._device_matrix(g)
        if m is None:
            raise ValueError('assemble_rows() fills a device-resident matrix')
        n_clients = len(clients_grads)
        if n_clients == 0:
            return
        n_seg = len(clients_grads[0])
        key = None
        if n_seg and _is_torch(clients_grads[0][0]):
            import torch
            try:     # C-level chain + map: no Python frame per tensor, no intermediate list
                first = clients_grads[0]
                # The table on the device is only valid for THESE tensors: the addresses of all of them, and -- because the
                # caching allocator can hand the same addresses back to tensors of another dtype or another split -- the
                # element counts, dtypes and contiguity of one client's tensors (every client has the same sizes: checked when
                # the table was built).  The fast path also skips the stream ordering of the general path, so it is only
                # taken when the launch stream IS torch's current stream.
                key = (n_clients, n_seg, m.cols, int(m.stream or 0),
                       int(torch.cuda.current_stream(first[0].device).cuda_stream),
                       tuple((t.numel(), t.dtype, t.is_contiguous()) for t in first),
                       tuple(map(torch.Tensor.data_ptr, itertools.chain.from_iterable(clients_grads))))
                if key[3] != key[4]:
                    key = None
            except (TypeError, AttributeError):
                key = None      # DeviceBuffers among them: the general path below
        if key is not None and key == self._assemble_key:      # (equal keys: equal tensor counts too)
            _check(self.lib.byz_assemble_rows_again_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, int(first_row), n_clients,
                                                        n_seg, _vp(m.stream)))
            return
        self._assemble_key = None
        flat = list(itertools.chain.from_iterable(clients_grads))
        if len(flat) != n_clients * n_seg:
            raise ValueError('every client must hand over the same number of tensors')
        ptrs, keep, lens, temporaries = [], [], None, False
        for c in range(n_clients):
            mine = []
            for t in flat[c * n_seg:(c + 1) * n_seg]:
                if isinstance(t, DeviceBuffer):
                    assert t.dtype == np.float32
                    ptrs.append(t.ptr)
                    mine.append(int(np.prod(t.shape)))
                elif _is_torch(t) and t.is_cuda:
                    import torch
                    if t.dtype != torch.float32:
                        raise ValueError('gradient tensors must be float32')
                    if not t.is_contiguous():
                        t, temporaries = t.contiguous(), True
                    ptrs.append(t.data_ptr())
                    mine.append(t.numel())
                else:
                    raise ValueError('assemble_rows() takes device tensors; host vectors go through assemble_row()')
                keep.append(t)
            if lens is None:
                lens = mine
            elif mine != lens:
                raise ValueError('clients disagree on the parameter sizes: %r and %r' % (lens, mine))
        sync_after = self._order_after_torch(m, keep)
        table = (ctypes.c_void_p * len(ptrs))(*ptrs)
        lengths = (ctypes.c_int64 * n_seg)(*lens)
        _check(self.lib.byz_assemble_rows_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, int(first_row), n_clients, n_seg,
                                              table, lengths, _vp(m.stream)))
        if sync_after:
            self.synchronize(m.stream)
        if key is not None and not temporaries:
            self._assemble_key = key      # what the device table now holds

    def assemble_columns(self, g, batched_grads):
        """Every client at once: `batched_grads[s]` is the device tensor (n_clients, *shape_s) holding parameter
        s's gradient for all clients (a batched client step); `g[:, start_s:start_s + len_s]` := its rows."""
        m = self._device_matrix(g)
        if m is None:
            raise ValueError('assemble_columns() fills a device-resident matrix')
        ptrs, lens, keep = [], [], []
        for t in batched_grads:
            if isinstance(t, DeviceBuffer):
                assert t.dtype == np.float32 and t.shape[0] == m.rows
                ptrs.append(t.ptr)
                lens.append(int(np.prod(t.shape[1:])))
            elif _is_torch(t) and t.is_cuda:
                import torch
                if t.dtype != torch.float32 or t.shape[0] != m.rows:
                    raise ValueError('batched gradients must be float32 with one leading row per client')
                t = t if t.is_contiguous() else t.contiguous()
                ptrs.append(t.data_ptr())
                lens.append(t.numel() // m.rows)
            else:
                raise ValueError('batched gradients must be device tensors')
            keep.append(t)
        sync_after = self._order_after_torch(m, keep)
        table = (ctypes.c_void_p * len(ptrs))(*ptrs)
        lengths = (ctypes.c_int64 * len(lens))(*lens)
        _check(self.lib.byz_assemble_columns_dev(self.ctx, _vp(m.ptr), m.rows, m.cols, m.ld, len(ptrs), table, lengths,
                                                 _vp(m.stream)))
        if sync_after:
            self.synchronize(m.stream)

    # ---- timing ------------------------------------------------------------------------------
    def timing(self, on=True):
        _check(self.lib.byz_timing_enable(self.ctx, int(bool(on))))
        _check(self.lib.byz_timing_reset(self.ctx))

    def timing_read(self):
        out = {}
        for k, name in enumerate(_native.KERNELS):
            ms, calls = ctypes.c_double(0.0), ctypes.c_int64(0)
            _check(self.lib.byz_timing_read(self.ctx, k, ctypes.byref(ms), ctypes.byref(calls)))
            if calls.value:
                out[name] = {'total_ms': ms.value, 'launches': calls.value}
        return out

    def lane_exchange_selftest(self):
        buf = DeviceBuffer(self, (16 * 64,), np.int32)
        n = ctypes.c_int32(0)
        _check(self.lib.byz_selftest_lane_exchange_dev(self.ctx, _vp(buf.ptr), ctypes.byref(n), None))
        self.synchronize()
        return buf.numpy()[: n.value * 64].reshape(n.value, 64)


_default = None


def get_engine():
    """Process-wide engine on BYZ_DEVICE / LOCAL_RANK (default GPU 0); created on first use."""
    global _default
    if _default is None:
        _default = Engine()
    return _default
