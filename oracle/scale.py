"""The selection rules of the reference at sizes numpy cannot finish.  TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

`oracle/csrc/select_ref.c` restates defences.py:26-37 (krum's scoring loop) and :59-68 (bulyan's pick-and-remove
loop) in plain C with OpenMP over the rows: every pick re-walks every live row's sorted distance list and sums its
first `users_count - corrupted_count` live entries, left to right, in fp32 (`mode='faithful'`, the reference's own
arithmetic under numpy >= 2) or in fp64 (`mode='ideal'`, with the relative margin of every pick).  O(theta * N^2).

Pinned by tests/test_oracle_scale.py: equal to `oracle.faithful` / `oracle.ideal` (themselves pinned bit for bit
against the imported reference and its golden vectors) on every case both can run.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(_HERE, 'csrc', 'select_ref.c')
LIB = os.path.join(_HERE, 'libselect_ref.so')
_lib = None


def build(force=False):
    """gcc -O2 -fopenmp -shared; the .so is git-ignored and travels to the GPU box with the tree."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SOURCE):
        return LIB
    cmd = ['gcc', '-O2', '-fopenmp', '-fPIC', '-shared', '-std=c11', '-Wall', '-o', LIB, SOURCE, '-lm']
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError('gcc failed for the selection oracle:\n%s\n%s' % (proc.stdout, proc.stderr))
    return LIB


def _load():
    global _lib
    if _lib is None:
        build()
        # a container's CPU quota can be far below what nproc reports: spinning OpenMP barriers on oversubscribed cores
        # turn seconds into tens of minutes, so the team is capped and made to sleep at barriers
        os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
        lib = ctypes.CDLL(LIB)
        i64, vp = ctypes.c_int64, ctypes.c_void_p
        lib.ref_krum_pick.argtypes = [vp, ctypes.c_int, i64, i64, ctypes.c_int, vp, vp]
        lib.ref_bulyan_selection.argtypes = [vp, ctypes.c_int, i64, i64, ctypes.c_int, vp, vp]
        lib.ref_set_threads.argtypes = [ctypes.c_int]
        lib.ref_verify_picks.argtypes = [vp, ctypes.c_int, i64, i64, ctypes.c_int, vp, ctypes.c_int, vp, ctypes.c_int, vp, vp]
        lib.ref_replay_selection.argtypes = [vp, ctypes.c_int, i64, i64, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp]
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        lib.ref_set_threads(max(1, min(usable, int(os.environ.get('BYZ_ORACLE_THREADS', '32')))))
        _lib = lib
    return _lib


_MODES = {'faithful': 0, 'ideal': 1}


def set_threads(n):
    return _load().ref_set_threads(int(n))


def _dense(dist):
    d = np.ascontiguousarray(dist, dtype=np.float32)
    assert d.ndim == 2 and d.shape[0] == d.shape[1]
    return d


def krum_pick(dist, users_count, corrupted_count, mode='faithful', with_scores=False):
    """defences.py:23-42 with return_index=True on a dense distance matrix (diagonal ignored)."""
    d = _dense(dist)
    n = d.shape[0]
    scores = np.empty(n, dtype=np.float64)
    margin = ctypes.c_double(0.0)
    idx = _load().ref_krum_pick(d.ctypes.data, n, int(users_count), int(corrupted_count), _MODES[mode],
                                scores.ctypes.data, ctypes.addressof(margin))
    assert idx >= -1
    return (idx, margin.value, scores) if with_scores else idx


def bulyan_selection(dist, users_count, corrupted_count, mode='faithful', with_margins=False):
    """The while loop of defences.py:59-68: indices in selection order (KeyError(-1) like the reference when a
    pick finds no score below 1e20)."""
    d = _dense(dist)
    n = d.shape[0]
    theta = int(users_count) - 2 * int(corrupted_count)
    sel = np.full(max(theta, 1), -1, dtype=np.int32)
    margins = np.zeros(max(theta, 1), dtype=np.float64)
    made = _load().ref_bulyan_selection(d.ctypes.data, n, int(users_count), int(corrupted_count), _MODES[mode],
                                        sel.ctypes.data, margins.ctypes.data)
    assert made >= -1
    if made < theta:
        raise KeyError(-1)
    picked = sel[:theta].tolist()
    return (picked, margins[:theta]) if with_margins else picked


def verify_picks(dist, users_count, corrupted_count, selection, picks, mode='faithful'):
    """For every pick index in `picks`: remove selection[:t], run the reference's scoring pass over the rest
    (defences.py:26-37) and compare its winner with selection[t].  Returns (mismatches, first_bad_pick, expected_row).
    One pick costs O(N^2); a whole selection O(theta N^2) -- use bulyan_selection for that."""
    d = _dense(dist)
    sel = np.ascontiguousarray(selection, dtype=np.int32)
    picks = np.ascontiguousarray(picks, dtype=np.int32)
    first_bad, expected = ctypes.c_int32(-1), ctypes.c_int32(-1)
    bad = _load().ref_verify_picks(d.ctypes.data, d.shape[0], int(users_count), int(corrupted_count), _MODES[mode],
                                   sel.ctypes.data, len(sel), picks.ctypes.data, len(picks),
                                   ctypes.addressof(first_bad), ctypes.addressof(expected))
    assert bad >= 0
    return bad, first_bad.value, expected.value


def replay_selection(dist, users_count, corrupted_count, selection, mode='ideal'):
    """SURVEY.md 8(d), the margin protocol's second clause.  `selection` was made on other numbers (the engine's distances
    and fp32 sums); here it is replayed on `dist` in `mode` arithmetic IN ITS OWN STATE: before pick t the rows selection[:t]
    are gone, every live row is scored (defences.py:26-37) and

        excess[t]  = score(selection[t]) / min score - 1     (0 where the selection's pick is an argmin of these scores)
        margin[t]  = (runner-up - min) / min                  (how contested pick t is)
        argmin[t]  = the row the rule picks in that state

    Returns (excess, margin, argmin).  The protocol: excess[t] <= tau for EVERY pick (also the ones after the first contested
    pick, where the selection and the oracle's own have parted ways), and the number of picks with margin[t] <= tau is
    reported."""
    d = _dense(dist)
    sel = np.ascontiguousarray(selection, dtype=np.int32)
    theta = len(sel)
    excess, margin = np.zeros(theta, dtype=np.float64), np.zeros(theta, dtype=np.float64)
    argmin = np.full(theta, -1, dtype=np.int32)
    rc = _load().ref_replay_selection(d.ctypes.data, d.shape[0], int(users_count), int(corrupted_count), _MODES[mode],
                                      sel.ctypes.data, theta, excess.ctypes.data, margin.ctypes.data, argmin.ctypes.data)
    if rc == -3:
        raise ValueError('replay_selection: the selection repeats a row or names one outside the matrix')
    assert rc == theta, rc
    return excess, margin, argmin
