"""Multi-GPU through the C ABI alone (needs an MI355X): byz_*_sharded_dev with the HOST's all-reduce as a callback
(include/byzagg.h, "multi-GPU, columns layout"; SURVEY.md 8(b) asked for communicators behind the boundary, 8(e) for the
columns layout as the cheaper equivalent of client sharding).

What a non-Python host would do with one thread (or process) per GPU and ncclAllReduce is done here on ONE GPU:

  * two ranks = two contexts on device 0 driven by two threads, each holding its own (uneven) slice of the columns; the
    callback IS an all-reduce -- both ranks' buffers are summed in fp64 and written back -- so the path under test is the
    one W = 2 runs: Gram of the slice -> all-reduce -> distances -> near-duplicate pairs over the local columns -> all-reduce
    -> apply -> replicated selection -> second stage on the local columns.  Checked against the oracle on the whole matrix
    (distances 1e-5; the selection exactly the reference's loop on the engine's own distance matrix -- two scores that
    differ in the last bits of a distance may swap two picks against fp32 `np.linalg.norm` distances, tests/test_gpu_scale.py's
    margin protocol is where that is pinned --; the aggregate 1e-5 on the selected rows: north_star's tolerance) and against
    the single-GPU entry points;
  * one rank with RCCL itself behind the callback (ncclCommInitRank at world size 1, ncclAllReduce on the call's stream):
    what INTEGRATION.md's C snippet does, through ctypes;
  * a callback that fails -> BYZ_E_COLLECTIVE with the callback's code in the text.

The inputs carry the attack's identical rows (malicious.py:26-27) and an honest pair that nearly coincides (defences.py:20
resolves it, the Gram identity does not), so both exchanges of the path happen.
"""
import ctypes
import os
import subprocess
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def attacked_matrix(n, d, f, seed):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if f:
        head = g[:f]
        g[:f] = (head.mean(axis=0) - 1.5 * head.std(axis=0)).astype(np.float32)
    g[f + 3] = g[f + 7] + np.float32(1e-4) * rng.standard_normal(d).astype(np.float32)    # a near-duplicate honest pair
    return g


class Rank:
    """One rank of the columns layout: its own context, its own slice of the columns on the device."""

    def __init__(self, g_slice):
        from attacking_federate_learning_amd.engine import Engine
        self.eng = Engine(0)
        self.n, self.d = g_slice.shape
        self.g = self.eng.to_device(g_slice)

    def close(self):
        self.g.free()
        self.eng.close()


def make_callback(fn):
    from attacking_federate_learning_amd import _native
    return _native.ALLREDUCE_F64_FN(fn)


def identity_allreduce(user, buf, count, stream):
    return 0


def bulyan_sharded(rank, users, f, cb, want_selection=True):
    from attacking_federate_learning_amd.engine import _check, _vp
    theta = users - 2 * f
    out = rank.eng.empty((rank.d,), np.float32)
    sel = rank.eng.empty((theta,), np.int32)
    _check(rank.eng.lib.byz_bulyan_sharded_dev(rank.eng.ctx, _vp(rank.g.ptr), rank.n, rank.d, rank.d, users, f,
                                               ctypes.cast(cb, ctypes.c_void_p), None, _vp(out.ptr),
                                               _vp(sel.ptr) if want_selection else None, None))
    rank.eng.check()
    return out.numpy(), sel.numpy()


def krum_sharded(rank, users, f, cb):
    from attacking_federate_learning_amd.engine import _check, _vp
    out = rank.eng.empty((rank.d,), np.float32)
    idx = ctypes.c_int32(-2)
    _check(rank.eng.lib.byz_krum_sharded_dev(rank.eng.ctx, _vp(rank.g.ptr), rank.n, rank.d, rank.d, users, f, 1,
                                             ctypes.cast(cb, ctypes.c_void_p), None, _vp(out.ptr), ctypes.byref(idx), None))
    return out.numpy(), int(idx.value)


def distances_sharded(rank, cb):
    from attacking_federate_learning_amd.engine import _check, _vp
    dist = rank.eng.empty((rank.n, rank.n), np.float32)
    _check(rank.eng.lib.byz_pairwise_distances_sharded_dev(rank.eng.ctx, _vp(rank.g.ptr), rank.n, rank.d, rank.d,
                                                           ctypes.cast(cb, ctypes.c_void_p), None, _vp(dist.ptr), None))
    rank.eng.check()
    return dist.numpy()


class TwoRankAllReduce:
    """An in-place SUM all-reduce over two ranks that live in two threads of this process: every rank downloads its buffer,
    the sums are formed on the host in fp64 in rank order (so both ranks get the same bits, as ncclAllReduce promises) and
    uploaded again.  `calls` records (count) per collective: both ranks must have made the same sequence."""

    def __init__(self, ranks):
        self.ranks = ranks
        self.barrier = threading.Barrier(len(ranks), timeout=120)
        self.staged = [None] * len(ranks)
        self.calls = [[] for _ in ranks]

    def callback_for(self, r):
        def allreduce(user, buf, count, stream):
            from attacking_federate_learning_amd.engine import _vp
            try:
                eng = self.ranks[r].eng
                host = np.empty(count, dtype=np.float64)
                if eng.lib.byz_download(eng.ctx, host.ctypes.data_as(ctypes.c_void_p), _vp(buf), 8 * count, _vp(stream)) != 0:
                    return 11
                self.staged[r] = host
                self.calls[r].append(int(count))
                self.barrier.wait()
                total = self.staged[0].copy()
                for other in self.staged[1:]:
                    total += other
                self.barrier.wait()            # everybody has read every staged buffer
                if eng.lib.byz_upload(eng.ctx, _vp(buf), total.ctypes.data_as(ctypes.c_void_p), 8 * count, _vp(stream)) != 0:
                    return 12
                if eng.lib.byz_stream_sync(eng.ctx, _vp(stream)) != 0:
                    return 13
                return 0
            except Exception:      # noqa: BLE001  (a Python exception must not unwind through the C frames)
                self.barrier.abort()
                return 99
        return make_callback(allreduce)


def run_ranks(ranks, work):
    """work(r, rank) on one thread per rank; returns the results in rank order, re-raises the first failure."""
    results, errors = [None] * len(ranks), [None] * len(ranks)

    def body(r):
        try:
            results[r] = work(r, ranks[r])
        except BaseException as exc:      # noqa: BLE001
            errors[r] = exc

    threads = [threading.Thread(target=body, args=(r,)) for r in range(len(ranks))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    for e in errors:
        if e is not None:
            raise e
    return results


@pytest.mark.parametrize('n,d,f,cut', [(200, 3000, 40, 1100), (131, 20000, 30, 16384), (64, 1500, 12, 700)])
def test_two_ranks_through_the_c_abi_match_the_oracle(eng, n, d, f, cut):
    from oracle import faithful
    g = attacked_matrix(n, d, f, seed=n + d)
    ranks = [Rank(g[:, :cut]), Rank(g[:, cut:])]
    try:
        # the distance matrix: identical on the two ranks, the oracle's to 1e-5, exact zeros among the attack's rows
        ar3 = TwoRankAllReduce(ranks)
        cbs3 = [ar3.callback_for(r) for r in range(2)]
        dists = run_ranks(ranks, lambda r, rank: distances_sharded(rank, cbs3[r]))
        assert np.array_equal(dists[0], dists[1])
        want = faithful.distance_matrix(g)
        off = ~np.eye(n, dtype=bool)
        assert np.allclose(dists[0][off], np.asarray(want, dtype=np.float32)[off], rtol=1e-5, atol=1e-6)
        assert np.all(dists[0][:f, :f][~np.eye(f, dtype=bool)] == 0.0)
        assert np.all(np.isinf(np.diag(dists[0])))
        # Bulyan: the reference's loop on these distances, pick for pick, on both ranks; the aggregate on those rows
        want_sel = faithful.bulyan_selection(dists[0], n, f)
        ar = TwoRankAllReduce(ranks)
        cbs = [ar.callback_for(r) for r in range(2)]
        res = run_ranks(ranks, lambda r, rank: bulyan_sharded(rank, n, f, cbs[r]))
        assert ar.calls[0] == ar.calls[1] and ar.calls[0][0] == n * n
        assert len(ar.calls[0]) == 2 and 1 <= ar.calls[0][1] <= n * n      # the Gram, then the near-duplicate pairs' list
        for r in range(2):
            assert np.array_equal(res[r][1], np.asarray(want_sel, dtype=np.int32))
        out = np.concatenate([res[0][0], res[1][0]])
        want_out = faithful.trimmed_mean(g[want_sel], len(want_sel), 2 * f)
        assert np.allclose(out, want_out, rtol=1e-5, atol=1e-5)
        # Krum over the same slices: the reference's index on every rank, each rank's columns of that row
        want_idx = faithful.krum(g, n, f, distances=dists[0], return_index=True)
        ar2 = TwoRankAllReduce(ranks)
        cbs2 = [ar2.callback_for(r) for r in range(2)]
        res2 = run_ranks(ranks, lambda r, rank: krum_sharded(rank, n, f, cbs2[r]))
        assert res2[0][1] == res2[1][1] == want_idx
        assert np.array_equal(np.concatenate([res2[0][0], res2[1][0]]), g[want_idx])
    finally:
        for rank in ranks:
            rank.close()


def test_one_rank_with_an_identity_all_reduce_is_the_single_gpu_call(eng):
    from attacking_federate_learning_amd.engine import _check, _vp
    n, d, f = 300, 5000, 60
    g = attacked_matrix(n, d, f, seed=5)
    rank = Rank(g)
    try:
        cb = make_callback(identity_allreduce)
        out, sel = bulyan_sharded(rank, n, f, cb)
        ref_out = rank.eng.empty((d,), np.float32)
        ref_sel = rank.eng.empty((n - 2 * f,), np.int32)
        _check(rank.eng.lib.byz_bulyan_dev(rank.eng.ctx, _vp(rank.g.ptr), n, d, d, n, f, _vp(ref_out.ptr), _vp(ref_sel.ptr), None))
        assert np.array_equal(sel, ref_sel.numpy())
        assert np.array_equal(out, ref_out.numpy())     # the same kernels on the same Gram: the same bits
    finally:
        rank.close()


def load_rccl():
    for name in ('librccl.so', 'librccl.so.1', '/opt/rocm/lib/librccl.so'):
        try:
            return ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    return None


def test_rccl_behind_the_callback_at_world_size_one(eng):
    """ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream) as the callback: INTEGRATION.md's C snippet."""
    rccl = load_rccl()
    if rccl is None:
        pytest.skip('librccl.so not loadable')

    class UniqueId(ctypes.Structure):
        _fields_ = [('internal', ctypes.c_char * 128)]

    n, d, f = 150, 4000, 30
    g = attacked_matrix(n, d, f, seed=9)
    rank = Rank(g)
    comm = ctypes.c_void_p()
    try:
        rank.eng.synchronize()       # (the context's device is current on this thread from here on)
        uid = UniqueId()
        rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
        rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        rccl.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p]
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
        assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
        counts = []

        def allreduce(user, buf, count, stream):
            counts.append(int(count))
            return rccl.ncclAllReduce(buf, buf, count, 8, 0, comm, stream)      # ncclDouble = 8 (ncclFloat64), ncclSum = 0

        out, sel = bulyan_sharded(rank, n, f, make_callback(allreduce))
        assert counts and counts[0] == n * n
        ref_out, ref_sel = bulyan_sharded(rank, n, f, make_callback(identity_allreduce))
        assert np.array_equal(sel, ref_sel) and np.array_equal(out, ref_out)      # a sum over one rank changes nothing
        from oracle import faithful
        dist = distances_sharded(rank, make_callback(allreduce))
        want_sel = faithful.bulyan_selection(dist, n, f)
        assert np.array_equal(sel, np.asarray(want_sel, dtype=np.int32))
        assert np.allclose(out, faithful.trimmed_mean(g[want_sel], len(want_sel), 2 * f), rtol=1e-5, atol=1e-5)
        assert np.allclose(dist[~np.eye(n, dtype=bool)], faithful.distance_matrix(g)[~np.eye(n, dtype=bool)], rtol=1e-5, atol=1e-6)
    finally:
        if comm.value:
            rccl.ncclCommDestroy(comm)
        rank.close()


def test_a_failing_all_reduce_is_reported(eng):
    from attacking_federate_learning_amd import _native
    from attacking_federate_learning_amd.engine import _vp
    g = attacked_matrix(140, 2000, 20, seed=3)
    rank = Rank(g)
    try:
        cb = make_callback(lambda user, buf, count, stream: 7)
        out = rank.eng.empty((rank.d,), np.float32)
        rc = rank.eng.lib.byz_bulyan_sharded_dev(rank.eng.ctx, _vp(rank.g.ptr), rank.n, rank.d, rank.d, 140, 20,
                                                 ctypes.cast(cb, ctypes.c_void_p), None, _vp(out.ptr), None, None)
        assert rc == _native.E_COLLECTIVE
        assert 'all-reduce returned 7' in _native.last_error()
        rc = rank.eng.lib.byz_bulyan_sharded_dev(rank.eng.ctx, _vp(rank.g.ptr), rank.n, rank.d, rank.d, 140, 20,
                                                 None, None, _vp(out.ptr), None, None)
        assert rc == _native.E_INVALID
        rank.eng.synchronize()
    finally:
        rank.close()


def c_host_command(out_path):
    """The build line of examples/shard_columns.c (its header comment), or None where RCCL's header is not installed."""
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    if not os.path.isfile(os.path.join(rocm, 'include', 'rccl', 'rccl.h')):
        return None
    lib_dir = os.path.join(ROOT, 'attacking_federate_learning_amd')
    return ['gcc', '-O2', '-std=c99', '-Wall', '-Wextra', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'),
            '-I', os.path.join(rocm, 'include'), os.path.join(ROOT, 'examples', 'shard_columns.c'), '-o', out_path,
            '-L', lib_dir, '-lbyzagg', '-L', os.path.join(rocm, 'lib'), '-lrccl', '-lamdhip64', '-lpthread', '-lm',
            '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,' + os.path.join(rocm, 'lib')]


@pytest.mark.timeout(600)
def test_the_c_host_example_runs_a_round_without_python(eng, tmp_path):
    """examples/shard_columns.c: a C program over include/byzagg.h + RCCL (ncclCommInitAll, one thread per GPU, ncclAllReduce
    behind the callback) -- built with gcc and run with the one GPU there is: every check of the program itself (the sharded
    Krum index and Bulyan selection are the unsharded ones, the aggregate within 1e-5) must hold."""
    exe = str(tmp_path / 'shard_columns')
    cmd = c_host_command(exe)
    if cmd is None:
        pytest.skip('rccl/rccl.h not installed')
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert build.returncode == 0 and not build.stderr.strip(), build.stderr[-2000:]     # no warnings either
    for args in (['1', '200', '5000', '40'], ['1', '131', '20000', '30'], ['1', '700', '3000', '150']):
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=280)
        assert run.returncode == 0 and run.stdout.strip().endswith('OK'), (args, run.stdout[-1500:], run.stderr[-1500:])
