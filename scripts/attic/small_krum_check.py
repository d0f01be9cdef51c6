"""GPU probe for csrc/krum_small.hip (torch-free: numpy + ctypes, starts in a second).

    BYZ_KRUM_SMALL is toggled per call; every case runs through both paths and against the oracle:
      * pairwise distances vs fp64 numpy and vs the general path,
      * Krum index vs oracle/faithful.py (the reference's loop) on the GPU's own distances and on the data,
      * the returned row,
    then both paths are timed (whole rounds and per kernel).  Prints as it goes.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from attacking_federate_learning_amd import _native                 # noqa: E402
from attacking_federate_learning_amd.engine import Engine           # noqa: E402
from oracle import faithful                                          # noqa: E402

OUT = open(os.path.join(ROOT, 'gpurun_out', 'small_krum.txt'), 'w') if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else None


def say(*a):
    line = ' '.join(str(x) for x in a)
    print(line, flush=True)
    if OUT:
        OUT.write(line + '\n')
        OUT.flush()


def make(n, d, seed, family):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d)).astype(np.float32)
    g *= (1.0 + 0.5 * rng.permutation(n) / n).astype(np.float32)[:, None]
    if family == 'attack':          # the first f rows are one vector (malicious.py:26-27)
        f = max(2, n // 4)
        v = (g[:f].mean(axis=0) - 1.5 * g[:f].std(axis=0)).astype(np.float32)
        g[:f] = v
    elif family == 'near':          # two rows that nearly coincide, and one exact copy
        g[5] = g[3] + np.float32(1e-4) * rng.standard_normal(d).astype(np.float32)
        g[7] = g[2]
    elif family == 'tiny':          # magnitudes around 1e-6
        g *= np.float32(1e-6)
    elif family == 'mixed':         # rows on very different scales
        g *= (10.0 ** rng.integers(-6, 6, size=n)).astype(np.float32)[:, None]
    return g


def dist64(g):
    x = g.astype(np.float64)
    out = np.zeros((len(g), len(g)))
    for i in range(len(g)):
        diff = (g[i][None, :] - g).astype(np.float32).astype(np.float64)   # fp32 difference, as the reference forms it
        out[i] = np.sqrt((diff * diff).sum(axis=1))
    np.fill_diagonal(out, np.inf)
    return out


MODES = ('0', '1')   # general path, csrc/krum_small.hip


def one_case(eng, n, d, f, family, seed):
    g = make(n, d, seed, family)
    buf = eng.to_device(g)
    want = dist64(g)
    res = {}
    for mode in MODES:
        os.environ['BYZ_KRUM_SMALL'] = mode[0]
        t0 = time.time()
        dm = eng.pairwise_distances(buf).numpy()
        idx = eng.krum(buf, n, f, return_index=True)
        row = eng.krum(buf, n, f).numpy()
        eng.check()
        res[mode] = (dm, idx, row, time.time() - t0)
    ok = True
    for mode in MODES:
        dm, idx, row, _ = res[mode]
        off = ~np.eye(n, dtype=bool)
        with np.errstate(invalid='ignore', divide='ignore'):
            rel = np.abs(dm[off] - want[off]) / np.maximum(want[off], 1e-300)
        rel = np.where(want[off] == 0, np.abs(dm[off]), rel)
        diag_ok = bool(np.all(np.isinf(np.diag(dm))))
        sym = bool(np.array_equal(dm, dm.T))
        # the reference's loop on THIS distance matrix must give THIS index
        ref_idx = faithful.krum(g, n, f, distances=dm, return_index=True)
        row_ok = bool(np.array_equal(row, g[idx]))
        good = rel.max() < 1e-6 and diag_ok and sym and ref_idx == idx and row_ok
        ok = ok and good
        say('  mode', mode, 'max rel err %.2e' % rel.max(), 'diag', diag_ok, 'sym', sym, 'idx', idx, 'ref idx on own dist', ref_idx,
            'row', row_ok, 'OK' if good else 'FAIL')
    same_idx = len({res[m][1] for m in MODES}) == 1
    dmax = np.nanmax(np.abs(np.where(np.isinf(res['0'][0]), 0, res['0'][0]) - np.where(np.isinf(res['1'][0]), 0, res['1'][0])))
    say('  paths agree on the index:', same_idx, ' max |d0 - d1| = %.3e' % dmax)
    if family in ('attack', 'near'):
        dm = res['1'][0]
        z = int((dm == 0).sum())
        say('  zeros in the small-path matrix:', z)
    return ok and same_idx


def timing(eng, n, d, f, rounds=200):
    g = make(n, d, 77, 'scaled')
    buf = eng.to_device(g)
    for mode in ('0', '1'):
        os.environ['BYZ_KRUM_SMALL'] = mode[0]
        for _ in range(20):
            eng.krum(buf, n, f)
        eng.synchronize()
        t0 = time.perf_counter()
        outs = [eng.krum(buf, n, f) for _ in range(rounds)]
        eng.synchronize()
        dt = (time.perf_counter() - t0) / rounds
        del outs
        eng.lib.byz_timing_reset(eng.ctx)
        eng.lib.byz_timing_enable(eng.ctx, 1)
        for _ in range(50):
            eng.krum(buf, n, f)
        eng.synchronize()
        import ctypes
        parts = []
        for k, name in enumerate(_native.KERNELS):
            ms, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            eng.lib.byz_timing_read(eng.ctx, k, ctypes.byref(ms), ctypes.byref(cnt))
            if cnt.value:
                parts.append('%s %.1f us x%d' % (name, 1e3 * ms.value / 50, cnt.value // 50))
        eng.lib.byz_timing_enable(eng.ctx, 0)
        say('timing N=%d D=%d mode %s: %.1f us per round (host-paired, %d rounds) |' % (n, d, mode, dt * 1e6, rounds), '; '.join(parts))


def main():
    eng = Engine(0)
    cases = [(10, 204, 2, 'scaled'), (23, 2048, 5, 'scaled'), (33, 129, 8, 'scaled'), (64, 100, 10, 'scaled'),
             (5, 7, 1, 'scaled'), (100, 21840, 24, 'scaled'), (100, 79510, 24, 'scaled'), (128, 4099, 30, 'scaled'),
             (97, 8190, 24, 'attack'), (100, 79510, 24, 'attack'), (40, 5000, 9, 'near'), (100, 79510, 24, 'near'),
             (50, 3000, 12, 'tiny'), (50, 3000, 12, 'mixed'), (2, 300, 0, 'scaled'), (128, 128, 31, 'scaled')]
    bad = 0
    for i, (n, d, f, fam) in enumerate(cases):
        say('case', i, 'N=%d D=%d f=%d %s' % (n, d, f, fam))
        try:
            if not one_case(eng, n, d, f, fam, 100 + i):
                bad += 1
        except Exception as e:   # keep going: one visit has to tell as much as possible
            bad += 1
            say('  EXCEPTION', type(e).__name__, e)
            if bad >= 4 and i < 6:
                say('too many early failures: stopping the cases')
                break
    say('cases failed:', bad, 'of', len(cases))
    for n, d in ((100, 79510), (100, 21840), (100, 1000000), (128, 79510)):
        try:
            timing(eng, n, d, 24)
        except Exception as e:
            say('timing EXCEPTION', type(e).__name__, e)
    os.environ['BYZ_KRUM_SMALL_HELPERS'] = '0'
    try:
        say('helpers off:')
        one_case(eng, 100, 79510, 24, 'attack', 555)
    except Exception as e:
        say('  EXCEPTION', type(e).__name__, e)
    say('done')


if __name__ == '__main__':
    main()
