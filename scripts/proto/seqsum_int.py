"""Prototype of the exact integer evaluation of the reference's sequential fp32 sum (defences.py:33-34: Python's sum() over
np.float32 values = left-to-right additions, each rounded to nearest-even) in 512-entry passes, as the GPU's re-score does it
(csrc/select.hip).  Checked bit for bit against the literal chain on random and adversarial sequences.

While the running sum s = I * q (q = its ulp, 2^23 <= I < 2^24, or subnormal) stays inside its binade,
    fl(s + x) = (I + a + t) * q,   a = floor(x / q),  t = [rem > q/2] or [rem == q/2 and I + a odd]
so a pass needs: a, the class of the remainder, the parity of I in front of every tie (a prefix over xor / reset), a prefix
sum, and the first entry whose sum reaches 2^24 -- that one entry is added literally, and the pass restarts behind it with
the new q.
"""
import numpy as np


def decompose(bits):
    e = bits >> 23
    frac = bits & 0x7fffff
    if e == 0:
        return frac, -149
    return frac | 0x800000, e - 150


def compose(I, Eq):
    if I == 0:
        return 0
    if Eq == -149 and I < (1 << 23):
        return I
    assert (1 << 23) <= I < (1 << 24), (I, Eq)
    return ((Eq + 150) << 23) | (I & 0x7fffff)


def f32(bits):
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def bits_of(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def classify(xb, Eq):
    """(a, above, tie) of one entry under unit 2^Eq; a saturates at 2^24."""
    M, E = decompose(xb)
    sh = Eq - E
    if M == 0:
        return 0, 0, 0
    if sh <= 0:
        a = M << (-sh) if -sh < 8 else 1 << 24
        return min(a, 1 << 24), 0, 0
    if sh >= 25:
        return 0, 0, 0
    a = M >> sh
    rem = M & ((1 << sh) - 1)
    half = 1 << (sh - 1)
    return a, int(rem > half), int(rem == half)


def lane_pass(entries, s_bits, start, lanes=64, per_lane=8):
    """One pass over a batch (len(entries) == lanes * per_lane, uint32 bit patterns) from position `start`:
    returns (new s bits, None) if the batch was consumed, or (s bits after the literal crossing add, position + 1)."""
    I, Eq = decompose(s_bits)
    summaries = []
    for L in range(lanes):
        s0 = s1 = 0
        p0, p1 = 0, 1
        for j in range(per_lane):
            pos = L * per_lane + j
            a, above, tie = classify(int(entries[pos]), Eq) if pos >= start else (0, 0, 0)
            alpha = a & 1
            t0 = above | (tie & (p0 ^ alpha))
            t1 = above | (tie & (p1 ^ alpha))
            s0 += a + t0
            s1 += a + t1
            p0 = 0 if tie else p0 ^ alpha ^ above
            p1 = 0 if tie else p1 ^ alpha ^ above
        summaries.append((min(s0, 1 << 25), min(s1, 1 << 25), p0, p1))
    constm = sum((1 << L) for L in range(lanes) if summaries[L][2] == summaries[L][3])
    valm = sum((1 << L) for L in range(lanes) if summaries[L][2] == 1)
    pin = []
    for L in range(lanes):
        lower = (1 << L) - 1
        cm = constm & lower
        if cm == 0:
            p = (I & 1) ^ (bin(valm & lower).count('1') & 1)
        else:
            j = cm.bit_length() - 1
            after = lower & ~((2 << j) - 1)
            p = ((valm >> j) & 1) ^ (bin(valm & after).count('1') & 1)
        pin.append(p)
    mine = [summaries[L][1] if pin[L] else summaries[L][0] for L in range(lanes)]
    incl = np.cumsum(mine)
    limit = (1 << 24) - I
    cross = [L for L in range(lanes) if incl[L] >= limit]
    if not cross:
        return compose(I + int(incl[-1]), Eq), None
    Lc = cross[0]
    run = I + int(incl[Lc] - mine[Lc])
    p = pin[Lc]
    for j in range(per_lane):
        pos = Lc * per_lane + j
        a, above, tie = classify(int(entries[pos]), Eq) if pos >= start else (0, 0, 0)
        alpha = a & 1
        t = above | (tie & (p ^ alpha))
        if run + a + t >= (1 << 24):
            before = f32(compose(run, Eq))
            with np.errstate(over='ignore'):
                after = np.float32(before) + f32(int(entries[pos]))
            return bits_of(after), pos + 1
        run += a + t
        p = 0 if tie else p ^ alpha ^ above
    raise AssertionError('the crossing lane holds no crossing entry')


def seqsum_int(values, literal_head=64, batch=512):
    vals = np.asarray(values, dtype=np.float32)
    n = len(vals)
    s = np.float32(0.0)
    head = min(literal_head, n)
    for v in vals[:head]:
        s = np.float32(s + v)
    s_bits = bits_of(s)
    passes = 0
    pos0 = 0
    while pos0 < n:
        chunk = np.zeros(batch, dtype=np.float32)
        m = min(batch, n - pos0)
        chunk[:m] = vals[pos0:pos0 + m]
        entries = chunk.view(np.uint32)
        start = max(head - pos0, 0)
        while start is not None and start < batch:
            if not np.isfinite(f32(s_bits)):
                return f32(s_bits), passes
            s_bits, start = lane_pass(entries, s_bits, start)
            passes += 1
        pos0 += batch
    return f32(s_bits), passes


def literal(values):
    s = np.float32(0.0)
    for v in np.asarray(values, dtype=np.float32):
        s = np.float32(s + v)
    return s


def main():
    rng = np.random.default_rng(5)
    cases = []
    for n in (1, 2, 63, 64, 65, 511, 512, 513, 1500, 3040, 7600):
        d = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.5))
        cases.append(('sorted distances %d' % n, d))
        z = d.copy()
        z[rng.random(n) < 0.3] = 0.0           # removed columns: + 0.0
        cases.append(('with zeros %d' % n, z))
    cases.append(('all equal (ties everywhere)', np.full(5000, 1.25, dtype=np.float32)))
    cases.append(('all equal 0.1', np.full(4097, 0.1, dtype=np.float32)))
    cases.append(('halves of the ulp', np.concatenate([[2.0 ** 20] * 70, [2.0 ** -4] * 3000]).astype(np.float32)))
    cases.append(('exact power crossings', np.concatenate([[1.0] * 64, [0.5] * 128, [64.0] * 30, [2.0 ** -10] * 2000]).astype(np.float32)))
    cases.append(('big after small', np.concatenate([[1e-3] * 100, [1e3] * 100, [1e-3] * 1000, [1e9], [1.0] * 600]).astype(np.float32)))
    cases.append(('subnormals', np.concatenate([np.full(200, 1e-45), np.full(300, 3e-39), np.full(400, 2e-38)]).astype(np.float32)))
    cases.append(('mixed magnitudes', np.abs(rng.standard_normal(6000) * 10.0 ** rng.integers(-6, 6, 6000)).astype(np.float32)))
    cases.append(('overflow', np.concatenate([[1.0] * 70, [3e38] * 5, [1.0] * 100]).astype(np.float32)))
    lattice = (rng.integers(0, 1 << 12, 4000).astype(np.float32) * np.float32(2.0 ** -9))   # many exact ties
    cases.append(('lattice (frequent ties)', lattice))
    cases.append(('lattice sorted', np.sort(lattice)))
    bad = 0
    for name, v in cases:
        want = literal(v)
        got, passes = seqsum_int(v)
        ok = bits_of(want) == bits_of(got) or (np.isnan(want) and np.isnan(got))
        bad += not ok
        print('%-34s n=%5d literal %-14.9g integer %-14.9g passes %3d  %s' % (name, len(v), want, got, passes, 'ok' if ok else 'MISMATCH'))
    for trial in range(300):
        n = int(rng.integers(1, 1400))
        kind = trial % 4
        if kind == 0:
            v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32))
        elif kind == 1:
            v = rng.integers(0, 1 << int(rng.integers(2, 20)), n).astype(np.float32) * np.float32(2.0 ** -int(rng.integers(0, 16)))
        elif kind == 2:
            v = np.abs(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3, n)).astype(np.float32)
        else:
            v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32))
            v[rng.random(n) < 0.5] = 0.0
        want = literal(v)
        got, _ = seqsum_int(v)
        if bits_of(want) != bits_of(got):
            bad += 1
            print('random trial %d (kind %d, n %d): literal %r integer %r MISMATCH' % (trial, kind, n, want, got))
    print('mismatches:', bad)
    return bad


if __name__ == '__main__' and len(__import__('sys').argv) == 1:
    raise SystemExit(1 if main() else 0)


# ---------------------------------------------------------------------------------------------------------------------
# Four waves on one contender (csrc/select.hip, cooperative re-score): a super-batch of 2048 entries, wave w the 512 entries
# from 512 w, every wave's work done for an incoming parity of 0 and patched afterwards -- only the FIRST lane that holds a
# tie can see the parity in front of the wave (lanes in front of it hold no tie, lanes behind it start from its even I).
def wave_summary(entries, base_pos, Eq, start, lanes=64, per_lane=8):
    """What one wave publishes: totals under an incoming parity of 0 / 1, outgoing parity under 0 / 1, and its lane state."""
    lane = []
    for L in range(lanes):
        s0 = s1 = 0
        p0, p1 = 0, 1
        for j in range(per_lane):
            pos = base_pos + L * per_lane + j
            a, above, tie = classify(int(entries[pos]), Eq) if pos >= start else (0, 0, 0)
            alpha = a & 1
            s0 += a + (above | (tie & (p0 ^ alpha)))
            s1 += a + (above | (tie & (p1 ^ alpha)))
            p0 = 0 if tie else p0 ^ alpha ^ above
            p1 = 0 if tie else p1 ^ alpha ^ above
        lane.append((min(s0, 1 << 25), min(s1, 1 << 25), p0, p1))
    tiem = sum((1 << L) for L in range(lanes) if lane[L][2] == lane[L][3])
    valm = sum((1 << L) for L in range(lanes) if lane[L][2] == 1)

    def pin_of(L, pw):
        lower = (1 << L) - 1
        cm = tiem & lower
        if cm == 0:
            return pw ^ (bin(valm & lower).count('1') & 1)
        j = cm.bit_length() - 1
        after = lower & ~((2 << j) - 1)
        return ((valm >> j) & 1) ^ (bin(valm & after).count('1') & 1)

    mine0 = [lane[L][1] if pin_of(L, 0) else lane[L][0] for L in range(lanes)]
    T0 = sum(mine0)
    if tiem == 0:
        T1 = T0
        out = (T0 & 1, (T0 & 1) ^ 1)
        ft = None
    else:
        ft = (tiem & -tiem).bit_length() - 1
        m1 = lane[ft][1] if pin_of(ft, 1) else lane[ft][0]
        T1 = T0 - mine0[ft] + m1
        o = pin_of(lanes, 0)          # the parity behind the last lane: absolute once a tie lane exists
        out = (o, o)
    return dict(lane=lane, pin_of=pin_of, mine0=mine0, T=(min(T0, 1 << 27), min(T1, 1 << 27)), out=out, ft=ft)


def super_pass(entries, s_bits, start, waves=4, lanes=64, per_lane=8):
    I, Eq = decompose(s_bits)
    summ = [wave_summary(entries, w * lanes * per_lane, Eq, start) for w in range(waves)]
    p, acc, base, par = I & 1, 0, [], []
    for w in range(waves):
        base.append(acc)
        par.append(p)
        acc += summ[w]['T'][p]
        p = summ[w]['out'][p]
    if I + acc < (1 << 24):
        return compose(I + acc, Eq), None
    wc = next(w for w in range(waves) if I + base[w] + summ[w]['T'][par[w]] >= (1 << 24))
    sw, pw = summ[wc], par[wc]
    mine = list(sw['mine0'])
    if pw == 1 and sw['ft'] is not None:
        ft = sw['ft']
        mine[ft] = sw['lane'][ft][1] if sw['pin_of'](ft, 1) else sw['lane'][ft][0]
    incl = np.cumsum(mine)
    Lc = next(L for L in range(lanes) if I + base[wc] + incl[L] >= (1 << 24))
    run = I + base[wc] + int(incl[Lc] - mine[Lc])
    pcur = sw['pin_of'](Lc, pw)
    for j in range(per_lane):
        pos = wc * lanes * per_lane + Lc * per_lane + j
        a, above, tie = classify(int(entries[pos]), Eq) if pos >= start else (0, 0, 0)
        alpha = a & 1
        t = above | (tie & (pcur ^ alpha))
        if run + a + t >= (1 << 24):
            before = f32(compose(run, Eq))
            with np.errstate(over='ignore'):
                after = np.float32(before) + f32(int(entries[pos]))
            return bits_of(after), pos + 1
        run += a + t
        pcur = 0 if tie else pcur ^ alpha ^ above
    raise AssertionError('no crossing entry in the crossing lane')


def seqsum_coop(values, literal_head=64, batch=2048):
    vals = np.asarray(values, dtype=np.float32)
    n = len(vals)
    s = np.float32(0.0)
    head = min(literal_head, n)
    for v in vals[:head]:
        s = np.float32(s + v)
    s_bits = bits_of(s)
    pos0 = 0
    while pos0 < n:
        chunk = np.zeros(batch, dtype=np.float32)
        m = min(batch, n - pos0)
        chunk[:m] = vals[pos0:pos0 + m]
        entries = chunk.view(np.uint32)
        start = max(head - pos0, 0)
        while start is not None and start < batch:
            if not np.isfinite(f32(s_bits)):
                return f32(s_bits)
            s_bits, start = super_pass(entries, s_bits, start)
        pos0 += batch
    return f32(s_bits)


def main_coop():
    rng = np.random.default_rng(6)
    bad = 0
    for trial in range(120):
        n = int(rng.integers(1, 9000))
        kind = trial % 4
        if kind == 0:
            v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.3))
        elif kind == 1:
            v = rng.integers(0, 1 << int(rng.integers(2, 14)), n).astype(np.float32) * np.float32(2.0 ** -int(rng.integers(0, 12)))
        elif kind == 2:
            v = np.abs(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3, n)).astype(np.float32)
        else:
            v = np.sort(rng.integers(1 << 6, 1 << 12, n).astype(np.float32) * np.float32(2.0 ** -9))
            v[rng.random(n) < 0.3] = 0.0
        want = literal(v)
        got = seqsum_coop(v)
        if bits_of(want) != bits_of(got):
            bad += 1
            print('coop trial %d (kind %d, n %d): literal %r four-wave %r MISMATCH' % (trial, kind, n, want, got))
    print('four-wave mismatches:', bad)
    return bad


if __name__ == '__main__' and len(__import__('sys').argv) > 1 and __import__('sys').argv[1] == 'coop':
    raise SystemExit(1 if main_coop() else 0)
