"""Prototype of the PARALLEL form of the exact sequential fp32 sum (round 4; its GPU form -- a cooperative re-score by all four waves of a workgroup -- was built, exact, and removed: profiles/r04r_bulyan_cooperative_rescore.txt).

The integer passes of seqsum_int.py are sequential only because a batch needs the unit (binade) of the running sum it starts
from.  But the running sum is known in advance to within its rounding error: with E_j the exact sum of the first j entries
(fp64), the fp32 partial sum s_j of j non-negative terms satisfies |s_j - E_j| <= u j E_j (u = 2^-24).  So for every batch of
512 entries whose enclosure [E_b (1 - d), E_b+1 (1 + d)] lies inside ONE binade, the unit is known before anything has been
added, the batch's total increment under an incoming parity of 0 / 1 (ties) can be formed independently of every other batch
(`wave_summary` of seqsum_int.py), and what remains sequential is a chain of integer additions, one per batch.  Batches whose
enclosure touches a power of two -- the ~log2(m / 512) crossings, and anything odd -- go through the sequential passes as
before, with the exact running sum.  Checked bit for bit against the literal chain.
"""
import math
import sys

import numpy as np

from seqsum_int import bits_of, compose, decompose, f32, lane_pass, literal, wave_summary

U = 2.0 ** -24


def clean_unit(e_lo_count, e_lo, e_hi_count, e_hi):
    """Eq of the binade that holds every fp32 partial sum of a batch, or None if the enclosure does not fit one binade."""
    if not (e_lo > 0.0 and math.isfinite(e_hi)):
        return None
    lo = e_lo * (1.0 - 1.1 * U * (e_lo_count + 2))
    hi = e_hi * (1.0 + 1.1 * U * (e_hi_count + 2))
    if lo < 2.0 ** -120 or hi >= 2.0 ** 127:
        return None
    k_lo, k_hi = math.floor(math.log2(lo)), math.floor(math.log2(hi))
    # (log2 of a double is exact enough: a value within rounding of a power of two fails the test on one side or the other
    #  only if it is within 1e-16 of it, far inside the margin 1.1 u j)
    if k_lo != k_hi or not (2.0 ** k_lo <= lo and hi < 2.0 ** (k_lo + 1)):
        return None
    return k_lo - 23


def seqsum_pred(values, batch=512, stats=None):
    vals = np.asarray(values, dtype=np.float32)
    n = len(vals)
    n_b = -(-n // batch) if n else 0
    padded = np.zeros(n_b * batch, dtype=np.float32)
    padded[:n] = vals
    entries = padded.view(np.uint32)
    # phase 1: exact sums per batch (fp64), their prefix
    sums = padded.astype(np.float64).reshape(n_b, batch).sum(axis=1) if n_b else np.zeros(0)
    start = np.concatenate([[0.0], np.cumsum(sums)])
    # phase 2 (parallel on the GPU): summaries of the clean batches under their predicted unit
    plan = []
    for b in range(n_b):
        eq = clean_unit(b * batch, start[b], (b + 1) * batch, start[b + 1]) if b > 0 else None
        if eq is None:
            plan.append(None)
        else:
            w = wave_summary(entries[b * batch:(b + 1) * batch], 0, eq, 0)
            plan.append((eq, w['T']))
    # the head: batch 0, literally
    s = np.float32(0.0)
    for v in padded[:batch] if n_b else []:
        s = np.float32(s + v)
    s_bits = bits_of(s)
    # phase 3: the chain
    seq = 0
    for b in range(1, n_b):
        if not np.isfinite(f32(s_bits)):
            break
        if plan[b] is not None:
            eq, T = plan[b]
            I, Eq = decompose(s_bits)
            if Eq == eq and (Eq > -149 or I >= (1 << 23)) and I + T[I & 1] < (1 << 24):
                s_bits = compose(I + T[I & 1], Eq)
                continue
            assert False, 'the enclosure said clean, the chain disagrees (b=%d Eq=%d predicted %d I=%d T=%r)' % (b, Eq, eq, I, T)
        seq += 1
        st = 0
        chunk = entries[b * batch:(b + 1) * batch]
        while st is not None and st < batch:
            if not np.isfinite(f32(s_bits)):
                break
            s_bits, st = lane_pass(chunk, s_bits, st)
    if stats is not None:
        stats.append((n_b, seq))
    return f32(s_bits)


def main():
    rng = np.random.default_rng(11)
    bad = 0
    stats = []
    cases = []
    for n in (1, 511, 512, 513, 1024, 1500, 3040, 7600, 12000):
        d = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.5))
        cases.append(('sorted distances %d' % n, d))
        z = d.copy()
        z[rng.random(n) < 0.3] = 0.0
        cases.append(('with zeros %d' % n, z))
    cases.append(('all equal (ties everywhere)', np.full(5000, 1.25, dtype=np.float32)))
    cases.append(('all equal 0.1', np.full(4097, 0.1, dtype=np.float32)))
    cases.append(('halves of the ulp', np.concatenate([[2.0 ** 20] * 70, [2.0 ** -4] * 3000]).astype(np.float32)))
    cases.append(('exact power crossings', np.concatenate([[1.0] * 64, [0.5] * 128, [64.0] * 30, [2.0 ** -10] * 2000]).astype(np.float32)))
    cases.append(('big after small', np.concatenate([[1e-3] * 100, [1e3] * 100, [1e-3] * 1000, [1e9], [1.0] * 600]).astype(np.float32)))
    cases.append(('subnormals', np.concatenate([np.full(200, 1e-45), np.full(300, 3e-39), np.full(400, 2e-38)]).astype(np.float32)))
    cases.append(('mixed magnitudes', np.abs(rng.standard_normal(6000) * 10.0 ** rng.integers(-6, 6, 6000)).astype(np.float32)))
    cases.append(('overflow', np.concatenate([[1.0] * 700, [3e38] * 5, [1.0] * 1000]).astype(np.float32)))
    lattice = (rng.integers(0, 1 << 12, 6000).astype(np.float32) * np.float32(2.0 ** -9))
    cases.append(('lattice (frequent ties)', lattice))
    cases.append(('lattice sorted', np.sort(lattice)))
    cases.append(('sum sits ON a power of two', np.concatenate([np.full(512, 1.0), np.full(512, 1.0), np.full(1024, 1.0), np.full(2048, 1.0)]).astype(np.float32)))
    cases.append(('sum just below a power of two', np.concatenate([np.full(1023, 1.0), [0.99999], np.full(3000, 1e-4)]).astype(np.float32)))
    for name, v in cases:
        want = literal(v)
        st = []
        got = seqsum_pred(v, stats=st)
        ok = bits_of(want) == bits_of(got) or (np.isnan(want) and np.isnan(got))
        bad += not ok
        print('%-34s n=%5d literal %-14.9g parallel %-14.9g batches %2d sequential %2d  %s' % (
            name, len(v), want, got, st[0][0], st[0][1], 'ok' if ok else 'MISMATCH'))
    for trial in range(400):
        n = int(rng.integers(1, 9000))
        kind = trial % 5
        if kind == 0:
            v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.3))
        elif kind == 1:
            v = rng.integers(0, 1 << int(rng.integers(2, 14)), n).astype(np.float32) * np.float32(2.0 ** -int(rng.integers(0, 12)))
        elif kind == 2:
            v = np.abs(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3, n)).astype(np.float32)
        elif kind == 3:
            v = np.sort(rng.integers(1 << 6, 1 << 12, n).astype(np.float32) * np.float32(2.0 ** -9))
            v[rng.random(n) < 0.3] = 0.0
        else:
            v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) * np.float32(10.0 ** rng.integers(-20, 20)))
        want = literal(v)
        got = seqsum_pred(v, stats=stats)
        if bits_of(want) != bits_of(got) and not (np.isnan(want) and np.isnan(got)):
            bad += 1
            print('random trial %d (kind %d, n %d): literal %r parallel %r MISMATCH' % (trial, kind, n, want, got))
    tot = sum(a for a, _ in stats)
    seq = sum(b for _, b in stats)
    print('mismatches: %d;  batches %d, of which sequential %d (%.0f%%)' % (bad, tot, seq, 100.0 * seq / max(tot, 1)))
    return bad


if __name__ == '__main__':
    raise SystemExit(1 if main() else 0)
