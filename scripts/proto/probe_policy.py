"""Prototype of the probe policy of the counting selection (how many counting passes until <= CAND candidates).
Not part of the product or the tests; it only tunes constants of csrc/median_window.hip."""
import numpy as np, sys

CAND = 16
NB = 0

def fkey(x):
    b = np.float32(x).view(np.uint32)
    return np.uint32(~b) if b & np.uint32(0x80000000) else np.uint32(b | np.uint32(0x80000000))

def from_fkey(k):
    k = np.uint32(k)
    b = np.uint32(k & np.uint32(0x7fffffff)) if k & np.uint32(0x80000000) else np.uint32(~k)
    return b.view(np.float32)

def select(vals, r, want_next, lo, hi, policy, delta=0.35, max_arith=12, rng=np.random.default_rng(7), thr=0.5, nbis=0):
    n = len(vals)
    c_lo, c_hi = 0, n
    r_hi = r + 1 if want_next else r
    probes = 0
    last = 0   # +1: last probe became hi, -1: became lo
    stall = 0
    same_side = 0
    w_lo = w_hi = 1.0
    prev_cand = n
    lo, hi = np.float32(lo), np.float32(hi)
    while c_hi - c_lo > CAND:
        klo, khi = int(fkey(lo)), int(fkey(hi))
        if khi - klo <= 1:
            return probes
        cand = c_hi - c_lo
        T = None
        f = None
        if probes < max_arith and np.isfinite(lo) and np.isfinite(hi):
            if policy == 'bisect' or (policy == 'interp' and stall >= 2) or probes < nbis:
                f = 0.5
                stall = 0
            elif policy in ('pivot', 'illinois') and stall >= 1:
                # data pivot (quickselect step): a candidate inside the bracket splits by rank, not by value
                inside = vals[(vals > lo) & (vals <= hi)]
                pv = np.float32(inside[rng.integers(len(inside))])
                # probing T = pv gives count >= its rank; use the float just below when pv == hi to stay inside
                f = None
                T = pv if pv < hi else from_fkey(int(fkey(pv)) - 1)
                if not (T > lo):
                    T = None
                stall = 0
            elif policy == 'illinois':
                want = r + 0.5 + (0.5 if want_next else 0.0)
                want += (-1 if last > 0 else 1) * delta * CAND
                g_lo, g_hi = (c_lo - want) * w_lo, (c_hi - want) * w_hi
                f = -g_lo / (g_hi - g_lo)
                f = min(max(f, 0.02), 0.98)
            else:
                want = r + 0.5 + (0.5 if want_next else 0.0)
                aim = want + (-1 if last > 0 else 1) * delta * CAND
                f = (aim - c_lo) / cand
                f = min(max(f, 0.02), 0.98)
            if f is not None:
                T = np.float32(np.float32(lo) + np.float32(f) * (np.float32(hi) - np.float32(lo)))
                if not (T > lo and T < hi):
                    T = None
        if T is None:
            T = from_fkey(klo + (khi - klo) // 2)
        c = int(np.count_nonzero(vals <= T))
        probes += 1
        if c <= r:
            same_side = same_side + 1 if last == -1 else 1
            if last == -1: w_hi *= 0.5
            else: w_hi = 1.0
            w_lo = 1.0
            lo, c_lo, last = T, c, -1
        elif c > r_hi:
            same_side = same_side + 1 if last == +1 else 1
            if last == +1: w_lo *= 0.5
            else: w_lo = 1.0
            w_hi = 1.0
            hi, c_hi, last = T, c, +1
        else:
            return probes + 1   # split pass
        new_cand = c_hi - c_lo
        stall = stall + 1 if new_cand > thr * cand else 0
    return probes

def run(name, gen, n, keepfrac, policy, trials=300, **kw):
    rng = np.random.default_rng(1)
    p1, p2 = [], []
    for _ in range(trials):
        x = gen(rng, n).astype(np.float32)
        mn, mx = x.min(), x.max()
        lo = from_fkey(int(fkey(mn)) - 1)
        if n & 1:
            p1.append(select(x, (n - 1) // 2, False, lo, mx, policy, **kw))
        else:
            p1.append(select(x, n // 2 - 1, True, lo, mx, policy, **kw))
        med = np.median(x).astype(np.float32)
        d = np.abs(x - med)
        keep = max(1, int(n * keepfrac))
        kw2 = dict(kw); kw2['nbis'] = NB
        p2.append(select(d, keep - 1, False, -np.float32(1e-45), d.max(), policy, **kw2))
    print('%-34s n=%5d keep=%.2f %-7s median %.2f (max %d)   t %.2f (max %d)' % (
        name, n, keepfrac, policy, np.mean(p1), max(p1), np.mean(p2), max(p2)))

gens = {
    'gauss': lambda rng, n: rng.standard_normal(n),
    'gauss+100': lambda rng, n: rng.standard_normal(n) + 100,
    'uniform': lambda rng, n: rng.random(n),
    'cauchy': lambda rng, n: rng.standard_cauchy(n),
    'attacked24%': lambda rng, n: np.concatenate([np.full(int(n * .24), -1.5), rng.standard_normal(n - int(n * .24))]),
    'outlier1e30': lambda rng, n: np.concatenate([[1e30, -1e30], rng.standard_normal(n - 2)]),
    'lognormal': lambda rng, n: np.exp(2 * rng.standard_normal(n)),
}
deltas = [0.35]
thrs = [0.85]
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for n, kf in ((1000, 0.799), (2080, 0.0764), (100, 0.79), (512, 0.8)):
    for name, g in gens.items():
        run(name, g, n, kf, 'bisect')
        for dl in deltas:
            run(name + ' d=%.2f' % dl, g, n, kf, 'interp', delta=dl)
            for th in thrs:
                run(name + ' thr=%.2f' % th, g, n, kf, 'pivot', delta=dl, thr=th)
    print()
