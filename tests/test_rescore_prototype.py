"""The Bulyan re-score's integer evaluation of the reference's sequential fp32 sum (csrc/select.hip: integer_passes) as its
Python prototype states it (scripts/proto/seqsum_int.py), against numpy's literal chain -- the CPU side of
tests/test_gpu_scale.py::test_bulyan_rescore_integer_passes_are_the_literal_chain.  Python's sum() over np.float32 values
(defences.py:33-34) is a left-to-right chain of round-to-nearest-even additions; the prototype must reproduce its bits."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def proto():
    spec = importlib.util.spec_from_file_location('seqsum_int', os.path.join(ROOT, 'scripts', 'proto', 'seqsum_int.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cases():
    rng = np.random.default_rng(11)
    out = []
    for n in (1, 2, 64, 65, 512, 513, 1300, 3040):
        d = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.5))
        out.append(('sorted distances %d' % n, d))
        z = d.copy()
        z[rng.random(n) < 0.3] = 0.0           # removed columns add + 0.0
        out.append(('with zeros %d' % n, z))
    out.append(('all equal: a tie at every other step', np.full(2500, 1.25, dtype=np.float32)))
    out.append(('halves of the ulp', np.concatenate([[2.0 ** 20] * 70, [2.0 ** -4] * 1500]).astype(np.float32)))
    out.append(('exact power crossings', np.concatenate([[1.0] * 64, [0.5] * 128, [64.0] * 30, [2.0 ** -10] * 900]).astype(np.float32)))
    out.append(('big after small', np.concatenate([[1e-3] * 100, [1e3] * 100, [1e-3] * 500, [1e9], [1.0] * 300]).astype(np.float32)))
    out.append(('subnormals', np.concatenate([np.full(200, 1e-45), np.full(300, 3e-39), np.full(400, 2e-38)]).astype(np.float32)))
    out.append(('overflow', np.concatenate([[1.0] * 70, [3e38] * 5, [1.0] * 100]).astype(np.float32)))
    lattice = (rng.integers(0, 1 << 12, 2000).astype(np.float32) * np.float32(2.0 ** -9))
    out.append(('lattice', lattice))
    out.append(('lattice sorted', np.sort(lattice)))
    return out


@pytest.mark.parametrize('name,values', cases(), ids=[c[0] for c in cases()])
def test_integer_passes_reproduce_the_sequential_fp32_sum(proto, name, values):
    with np.errstate(over='ignore'):
        want = proto.literal(values)
        got, passes = proto.seqsum_int(values)
    assert proto.bits_of(want) == proto.bits_of(got), (name, want, got)
    assert passes <= 4 + len(values) // 512 + 24          # a pass per batch plus one per binade crossing


def test_four_wave_composition_reproduces_it_too(proto):
    """The patch-one-lane composition over four waves (measured slower on the GPU and not shipped, but it pins the parity
    algebra: only the first tie lane of a wave sees the parity in front of the wave)."""
    rng = np.random.default_rng(12)
    for trial in range(12):
        n = int(rng.integers(1, 5000))
        if trial % 2:
            v = np.sort(rng.integers(1 << 6, 1 << 12, n).astype(np.float32) * np.float32(2.0 ** -9))
            v[rng.random(n) < 0.3] = 0.0
        else:
            v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.3))
        assert proto.bits_of(proto.literal(v)) == proto.bits_of(proto.seqsum_coop(v))
