"""csrc/rescore_incr.hpp -- the exact incremental update of the reference's sequential fp32 score between two Bulyan picks --
compiled for the HOST with g++ and checked bit for bit against numpy's literal chain (defences.py:33-34: Python's sum() over
np.float32 values).  The same header is what the GPU loop runs wave-uniformly; scripts/proto/seqsum_incr.py is its model."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GONE = np.uint32(0x80000000)


@pytest.fixture(scope='module')
def lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('incr') / 'librescore_incr_host.so')
    subprocess.run(['g++', '-O1', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', '-Wall', '-Werror',
                    '-I', os.path.join(ROOT, 'attacking_federate_learning_amd', 'csrc'),
                    os.path.join(ROOT, 'tests', 'native', 'rescore_incr_host.cpp'), '-o', out], check=True)
    lib = ctypes.CDLL(out)
    lib.incr_literal.restype = ctypes.c_uint32
    lib.incr_literal.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.incr_full.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.incr_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return lib


def literal(vals, end):
    """numpy's own chain over the live entries (the -0.0 marks and +0.0 add nothing)."""
    s = np.float32(0.0)
    for x in vals[:end].view(np.float32):
        s = np.float32(s + x)
    return int(np.array([s]).view(np.uint32)[0])


class Rec:
    def __init__(self, lib):
        self.buf = np.zeros(lib.incr_record_bytes() // 4, dtype=np.uint32)

    ptr = property(lambda self: self.buf.ctypes.data)
    s = property(lambda self: int(self.buf[0]))
    head_end = property(lambda self: int(self.buf[2].view(np.int32)))
    end = property(lambda self: int(self.buf[3].view(np.int32)))
    valid = property(lambda self: int(self.buf[4].view(np.int32)) >= 0)
    n_events = property(lambda self: int(self.buf[5].view(np.int32)))

    def event_positions(self):
        return [int(self.buf[6 + 4 * i].view(np.int32)) for i in range(self.n_events)]


def cases():
    rng = np.random.default_rng(23)
    out = []
    for n in (40, 600, 1500, 3040, 7600):
        out.append(('distance-like %d' % n, np.sort((1.0 + 0.25 * rng.random(n)).astype(np.float32) * np.float32(37.0)), 512))
    out.append(('a short head', np.sort((1.0 + 0.25 * rng.random(2000)).astype(np.float32) * np.float32(0.37)), 64))
    out.append(('all equal: a tie at every other step', np.full(1800, 1.25, dtype=np.float32), 512))
    out.append(('lattice: ties everywhere', np.sort(rng.integers(1, 1 << 12, 2200).astype(np.float32) * np.float32(2.0 ** -9)), 512))
    out.append(('twins in front (the attack): zeros, then distances',
                np.concatenate([np.zeros(700, dtype=np.float32), np.sort(rng.random(1500).astype(np.float32) + np.float32(3.0))]), 700 + 512))
    out.append(('exact powers of two', np.concatenate([[1.0] * 600, [2.0] * 600, [4.0] * 600]).astype(np.float32), 512))
    out.append(('wide range', np.sort(np.exp(rng.uniform(-20.0, 20.0, 2000)).astype(np.float32)), 512))
    out.append(('subnormals first', np.sort(np.concatenate([np.full(300, 3e-39), rng.random(1500) + 0.5]).astype(np.float32)), 512))
    return out


@pytest.mark.parametrize('name,values,head', cases(), ids=[c[0] for c in cases()])
def test_incremental_update_is_the_literal_chain_after_every_pick(lib, name, values, head):
    """One entry leaves the prefix per pick: the winner's distance (marked with -0.0 where it lies) or, when the winner lay
    behind the prefix, the last live entry.  After every pick the record's sum must be the literal chain's, whether the
    update came from the events or fell back to the full chain; on distance-like rows it must nearly always come from the events."""
    rng = np.random.default_rng(len(values))
    vals = values.view(np.uint32).copy()
    rec = Rec(lib)
    end = len(vals) - len(vals) // 10
    lib.incr_full(vals.ctypes.data, end, head, rec.ptr)
    assert rec.s == literal(vals, end) == lib.incr_literal(vals.ctypes.data, end)
    from_events = fallbacks = 0
    for pick in range(min(300, len(vals) // 3)):
        live = np.flatnonzero(vals[:rec.end] != GONE)
        if len(live) < 2:
            break
        mode = rng.random()
        if mode < 0.6:
            k = int(live[min(len(live) - 1, int(rng.exponential(len(live) / 20.0)))])   # near the front, as winners are
        elif mode < 0.8:
            k = int(live[rng.integers(len(live))])                                        # anywhere in the prefix
        elif mode < 0.88 and rec.valid and rec.n_events:
            k = rec.event_positions()[int(rng.integers(rec.n_events))]                   # an event itself (a tie, a crossing)
            if vals[k] == GONE:
                k = -1
        else:
            k = -1                                                                        # the winner lay behind the prefix
        if k < 0:
            want_end = int(live[-1])
        else:
            want_end = rec.end
        got = lib.incr_update(vals.ctypes.data, rec.ptr, k)
        assert rec.end == want_end, '%s: pick %d (k = %d): end %d, expected %d' % (name, pick, k, rec.end, want_end)
        want = literal(vals, rec.end)
        assert rec.s == want, '%s: pick %d (k = %d, %s): %08x, literal %08x' % (
            name, pick, k, 'from the events' if got else 'full chain', rec.s, want)
        from_events += got
        fallbacks += 1 - got
    # (rows with more than 24 events behind their head -- a tie at every other step, a sum that doubles every few entries -- keep
    #  no record at all: every update of theirs is the full chain, which is what the assertion above checked)
    if name.startswith('distance-like') and len(values) > 1000:
        # (a twelfth of the picks above mark an event on purpose, and marking a crossing IS a fallback)
        assert fallbacks <= 0.2 * (from_events + fallbacks), (from_events, fallbacks)
