"""Prototype of the multi-segment form of the parallel sequential sum (the second version of the GPU's cooperative re-score; built, exact, removed: profiles/r04r_bulyan_cooperative_rescore.txt):
the head is the first NON-EMPTY batch; inside every later batch each entry whose partial sums before and after it lie in one
binade for sure belongs to that binade's segment (up to 4 segments per batch), the entries in between are windows (<= 8 each),
added literally by the chain."""
import math
import numpy as np
from seqsum_int import bits_of, compose, decompose, f32, lane_pass, literal, wave_summary
U = 2.0 ** -24
SEGS, WIN = 4, 8

def margin(t): return 1.1 * U * (t + 2)
def expo(v): return math.floor(math.log2(v))

def seqsum_seg(values, batch=512, stats=None):
    vals = np.asarray(values, dtype=np.float32); n = len(vals); nb = -(-n // batch) if n else 0
    pad = np.zeros(nb * batch, dtype=np.float32); pad[:n] = vals; ent = pad.view(np.uint32)
    sums = pad.astype(np.float64).reshape(nb, batch).sum(1) if nb else np.zeros(0)
    cnts = (pad.reshape(nb, batch) != 0).sum(1) if nb else np.zeros(0, dtype=int)
    start = np.concatenate([[0.0], np.cumsum(sums)])
    head = next((b for b in range(nb) if cnts[b] > 0), nb)
    s = np.float32(0.0)
    if head < nb:
        for v in pad[head * batch:(head + 1) * batch]: s = np.float32(s + v)
    sb = bits_of(s); st = dict(clean=0, split=0, seq=0, fail=0)
    for b in range(head + 1, nb):
        if not np.isfinite(f32(sb)): break
        chunk = ent[b * batch:(b + 1) * batch]; x = pad[b * batch:(b + 1) * batch].astype(np.float64)
        e0 = start[b]; plan = None
        if e0 > 0 and start[b + 1] < 1e38:
            lo = e0 * (1 - margin(batch * b)); hi = start[b + 1] * (1 + margin(batch * (b + 1)))
            if lo >= 2.0 ** -120 and hi < 2.0 ** 127 and expo(hi) - expo(lo) < SEGS:
                klo = expo(lo); m = margin(batch * (b + 1))
                run = e0 + np.concatenate([[0.0], np.cumsum(x)[:-1]]); after = run + x
                seg = np.full(batch, -1); winid = np.full(batch, -1); ok = True
                for j in range(batch):
                    if x[j] == 0: continue
                    a, c = expo(run[j] * (1 - m)), expo(after[j] * (1 + m))
                    if a == c: seg[j] = a - klo
                    elif c == a + 1: winid[j] = a - klo
                    else: ok = False
                if ok and all((winid == w).sum() <= WIN for w in range(SEGS - 1)):
                    plan = (klo, seg, winid)
        sin = sb; done = False
        if plan is not None:
            klo, seg, winid = plan; done = True; sbb = sb
            for k in range(SEGS):
                msk = seg == k
                if msk.any():
                    A = chunk.copy(); A[~msk] = 0
                    I, E = decompose(sbb); T = wave_summary(A, 0, klo + k - 23, 0)['T']
                    if not (E == klo + k - 23 and I + T[I & 1] < (1 << 24)): done = False; break
                    sbb = compose(I + T[I & 1], E)
                if k < SEGS - 1:
                    sf = f32(sbb)
                    for w in pad[b * batch:(b + 1) * batch][winid == k]: sf = np.float32(sf + w)
                    sbb = bits_of(sf)
            if done:
                sb = sbb; st['split' if (seg.max() > 0 or (winid >= 0).any()) else 'clean'] += 1
            else: st['fail'] += 1
        if not done:
            st['seq'] += 1; sb = sin; p = 0
            while p is not None and p < batch:
                if not np.isfinite(f32(sb)): break
                sb, p = lane_pass(chunk, sb, p)
    if stats is not None: stats.append(st)
    return f32(sb)

if __name__ == '__main__':
    rng = np.random.default_rng(21); bad = 0; tot = dict(clean=0, split=0, seq=0, fail=0)
    for t in range(150):
        n = int(rng.integers(600, 9000)); k = t % 5
        if k == 0: v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.3))
        elif k == 1:   # removals concentrated at the front (the winners' distances): sparse early batches
            v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.5)); gone = rng.random(n) < np.exp(-np.arange(n) / (0.3 * n)); v[gone] = 0
        elif k == 2: v = np.sort(rng.integers(1 << 6, 1 << 12, n).astype(np.float32) * np.float32(2.0 ** -9)); v[:int(0.4 * n)][rng.random(int(0.4 * n)) < 0.9] = 0
        elif k == 3: v = np.abs(rng.standard_normal(n) * 10.0 ** rng.integers(-3, 3, n)).astype(np.float32)
        else: v = np.sort(np.abs(rng.standard_normal(n)).astype(np.float32) + np.float32(0.5)); v[:int(rng.integers(0, n // 2))] = 0
        stt = []; got = seqsum_seg(v, stats=stt); want = literal(v)
        if bits_of(got) != bits_of(want) and not (np.isnan(got) and np.isnan(want)): bad += 1; print('MISMATCH trial', t, k, n)
        for kk in tot: tot[kk] += stt[0][kk]
    print('mismatches', bad, tot)
