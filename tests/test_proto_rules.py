"""The numpy replays of kernel-side arithmetic rules that DESIGN.md cites (scripts/proto/): they must stay true.

These are development aids, not product code: each restates, on the CPU, an exactness argument a kernel relies on (or, for
the tie rule, one that was measured on the GPU and recorded as not worth its instructions)."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'scripts', 'proto', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_integer_rule_with_ties_is_the_sequential_fp32_sum():
    """csrc/select.hip (reference_score): inside one binade the left-to-right fp32 sum of a 64-entry chunk is an integer
    sum; an entry exactly half way between two multiples of the ulp adds floor + parity and leaves the sum even."""
    rule = _load('rescore_rule')
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(1, 4000))
        kind = trial % 4
        if kind == 0:
            v = np.sqrt(rng.chisquare(16, n))
        elif kind == 1:
            v = rng.integers(1, 1 << 12, n) / 64.0            # quantised: ties in almost every chunk
        elif kind == 2:
            v = np.full(n, rng.uniform(0.1, 10.0))            # one value: structural ties
        else:
            v = np.exp(rng.uniform(-20, 20, n))               # many decades: binade crossings, entries beyond 2^24 ulps
        v = np.sort(v.astype(np.float32))
        take = int(rng.integers(1, n + 1))
        v[take:] = 0.0
        want = rule.sequential(v[:take])
        got, _, _ = rule.by_rule(v)
        assert np.float32(got).view(np.uint32) == want.view(np.uint32), (trial, kind, n, take)
