"""Prototype of an exact INCREMENTAL re-score for the Bulyan loop -- not built on the GPU.

The reference's Krum score is Python's sum() over np.float32 values in ascending order (defences.py:33-34): a chain of
round-to-nearest-even additions.  Between two picks of the Bulyan loop a row's prefix changes by ONE entry: the winner's
distance is marked (it adds nothing from then on), or -- if the winner lay behind the prefix -- the prefix loses its last
live entry.  Re-scoring the 7600-entry chain for that costs ~46,000 cycles on the GPU (csrc/select.hip).  This file shows
that the new sum follows from the old chain's EVENTS alone:

  while the running sum s = I q stays inside one binade (q its ulp, 2^23 <= I < 2^24),
      fl(s + x) = (I + a + t) q,   a = floor(x / q),   t = [rem > q/2] or [rem == q/2 and I + a odd],
  so the increment of an entry depends on the entry and the unit only -- not on the sum -- except
    * at a TIE (rem == q/2 exactly): it depends on the parity of I in front of it;
    * at a CROSSING (I + a + t >= 2^24): the addition is an fp32 addition into the next binade, unit 2q from there on.
  Marking entry k lowers every later partial sum by the same integer `shift` (= k's own increment) until the next event;
  a tie changes the shift by +-1 when the shift is odd; around a crossing the two chains are in different binades for an
  entry or two (the lower one crosses later) -- there both are added literally until they meet again, and the shift is
  read off anew.

`Record.full` is the literal chain that also notes its events (what a full re-score on the GPU would store: ~13 crossings
and a handful of ties per 7600-entry prefix, most of the crossings inside the first 512 entries, which an update simply
redoes).  `Record.mark` is the update: exact, or it reports that it fell back to the literal chain (the marked entry is a
crossing itself, the chains do not meet within 16 entries, ...).  Checked bit for bit against the literal chain in
tests/test_rescore_prototype.py; `python scripts/proto/seqsum_incr.py` prints how long the event walks are on distance-like
rows.
"""
import numpy as np

HEAD = 512          # physical positions [0, HEAD) are always re-added literally (the sum changes binade every few steps there)
MEET_LIMIT = 16     # literal steps a crossing region may take before the update gives up


def bits_of(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


def f32(bits):
    return np.array([bits], dtype=np.uint32).view(np.float32)[0]


def decompose(bits):
    """(I, Eq): value = I * 2^Eq with 2^23 <= I < 2^24 (normal) or I < 2^23, Eq = -149 (subnormal / zero)."""
    e = bits >> 23
    frac = bits & 0x7fffff
    if e == 0:
        return frac, -149
    return frac | 0x800000, e - 150


def compose(I, Eq):
    if Eq == -149 and I < (1 << 23):
        return I
    assert (1 << 23) <= I < (1 << 24), (I, Eq)
    return ((Eq + 150) << 23) | (I & 0x7fffff)


def classify(xb, Eq):
    """(a, above, tie) of one non-negative entry under the unit 2^Eq."""
    M, E = decompose(xb)
    if M == 0:
        return 0, 0, 0
    sh = Eq - E
    if sh <= 0:
        return M << (-sh), 0, 0
    if sh >= 26:
        return 0, 0, 0
    a = M >> sh
    rem = M & ((1 << sh) - 1)
    half = 1 << (sh - 1)
    return a, int(rem > half), int(rem == half)


def step(s_bits, xb):
    """One addition of the chain.  Returns (new bits, kind, t): kind '' (plain), 'tie' (t = how it was resolved) or 'cross'."""
    if xb == 0:
        return s_bits, '', 0
    new_bits = bits_of(np.float32(f32(s_bits) + f32(xb)))
    I, Eq = decompose(s_bits)
    a, above, tie = classify(xb, Eq)
    t = above | (tie & ((I + a) & 1))
    if I + a + t >= (1 << 24) or decompose(new_bits)[1] != Eq:
        return new_bits, 'cross', 0
    assert new_bits == compose(I + a + t, Eq)
    return new_bits, 'tie' if tie else '', t


class Record:
    """The chain over vals[0 .. end) (uint32 bit patterns of non-negative floats; 0 = marked / adds nothing) and its events
    behind the literal head."""

    def __init__(self, vals, end):
        self.vals = vals          # the row's table; `mark` writes zeros into it like the loop writes -0.0
        self.end = end            # physical end of the prefix
        self.full()

    # ---- the literal chain, noting its events (a full re-score)
    def full(self):
        s = 0
        self.events = []          # behind the head, ascending positions: [pos, 'tie', t] or [pos, 'cross', bits before, bits after]
        self.s_head = 0
        for p in range(self.end):
            if p == min(HEAD, self.end):
                self.s_head = s
            xb = int(self.vals[p])
            new, kind, t = step(s, xb)
            if p >= HEAD and kind == 'tie':
                self.events.append([p, 'tie', t])
            elif p >= HEAD and kind == 'cross':
                self.events.append([p, 'cross', s, new])
            s = new
        if self.end <= HEAD:
            self.s_head = s
        self.s = s
        return s

    def literal(self):
        s = np.float32(0.0)
        for p in range(self.end):
            s = np.float32(s + f32(int(self.vals[p])))
        return bits_of(s)

    def unit_at(self, p):
        """Eq of the running sum in front of position p >= HEAD (from the head's sum and the crossings before p)."""
        eq = decompose(self.s_head)[1]
        for ev in self.events:
            if ev[0] >= p:
                break
            if ev[1] == 'cross':
                eq = decompose(ev[3])[1]
        return eq

    # ---- the update: entry k is marked (k < end), or the prefix loses its last live entry (k is None)
    def mark(self, k):
        """Returns the number of table entries the update had to read (its cost), or -1 after a fallback to `full`."""
        if k is None:
            return self._drop_last()
        xk = int(self.vals[k])
        assert k < self.end and xk != 0
        reads = 1
        if k < HEAD:
            self.vals[k] = 0
            s_new = 0
            for p in range(min(HEAD, self.end)):
                s_new, _, _ = step(s_new, int(self.vals[p]))
            reads += min(HEAD, self.end)
            s_old = self.s_head
            self.s_head = s_new
            if self.end <= HEAD:
                self.s = s_new
                return reads
            at = HEAD                       # both chains stand in front of position `at`
            first_event = 0
        else:
            hit = [i for i, ev in enumerate(self.events) if ev[0] == k]
            if hit and self.events[hit[0]][1] == 'cross':
                self.vals[k] = 0
                self.full()
                return -1
            eq = self.unit_at(k)
            a, above, tie = classify(xk, eq)
            t = self.events[hit[0]][2] if hit else above
            assert not tie or hit
            if hit:
                del self.events[hit[0]]
            self.vals[k] = 0
            shift, shift_eq = a + t, eq
            first_event = next((i for i, ev in enumerate(self.events) if ev[0] > k), len(self.events))
            at = None
        # the walk.  Either (shift, shift_eq): the new chain is `shift` units of 2^shift_eq below the old one and in the same
        # binade; or (s_new, s_old, at): both chains explicitly in front of position `at` (head redone, or a crossing region)
        i = first_event
        while True:
            if at is not None:
                # literal steps until both chains are in the same binade again (and not at a recorded event)
                steps = 0
                fresh = []
                while decompose(s_new)[1] != decompose(s_old)[1] or (i < len(self.events) and self.events[i][0] == at and self.events[i][1] == 'cross'):
                    if at >= self.end or steps >= MEET_LIMIT:
                        if at >= self.end and steps < MEET_LIMIT:
                            break
                        self.full()
                        return -1
                    xb = int(self.vals[at])
                    before = s_new
                    s_new, kind, t = step(s_new, xb)
                    s_old, _, _ = step(s_old, xb)
                    if kind == 'tie':
                        fresh.append([at, 'tie', t])
                    elif kind == 'cross':
                        fresh.append([at, 'cross', before, s_new])
                    reads += 1
                    steps += 1
                    at += 1
                    while i < len(self.events) and self.events[i][0] < at:   # the old chain's events inside the region are replaced
                        del self.events[i]
                self.events[i:i] = fresh
                i += len(fresh)
                if at >= self.end:
                    self.s = s_new
                    return reads
                (I_new, eq), (I_old, _) = decompose(s_new), decompose(s_old)
                shift, shift_eq = I_old - I_new, eq
                assert shift >= 0
                at = None
            if i >= len(self.events):
                break
            ev = self.events[i]
            if ev[1] == 'tie':
                if shift & 1:
                    new_t = ev[2] ^ 1
                    shift += ev[2] - new_t
                    ev[2] = new_t
                i += 1
                continue
            # a crossing of the old chain at ev[0]: the new chain stands `shift` units lower in front of it
            I_before, eq_before = decompose(ev[2])
            assert eq_before == shift_eq
            if I_before - shift < (1 << 23):
                self.full()
                return -1
            s_new, s_old, at = compose(I_before - shift, eq_before), ev[2], ev[0]
        I_fin, eq_fin = decompose(self.s)
        assert eq_fin == shift_eq
        if I_fin - shift < (1 << 23) and not (eq_fin == -149):
            self.full()
            return -1
        self.s = compose(I_fin - shift, eq_fin)
        return reads

    def _drop_last(self):
        p = self.end - 1
        while p >= 0 and int(self.vals[p]) == 0:
            p -= 1
        if p < 0:
            self.end = 0
            self.full()
            return 1
        if p < HEAD:
            self.end = p
            self.full()
            return -1
        xb = int(self.vals[p])
        reads = 1
        if self.events and self.events[-1][0] == p:
            ev = self.events.pop()
            if ev[1] == 'cross':
                self.s = ev[2]
            else:
                I, eq = decompose(self.s)
                a, above, tie = classify(xb, eq)
                I -= a + ev[2]
                if I < (1 << 23):
                    self.end = p
                    self.full()
                    return -1
                self.s = compose(I, eq)
        else:
            I, eq = decompose(self.s)
            a, above, tie = classify(xb, eq)
            assert not tie
            I -= a + above
            if I < (1 << 23):
                self.end = p
                self.full()
                return -1
            self.s = compose(I, eq)
        self.end = p
        return reads


def distance_like_row(rng, n, spread=0.25):
    """Ascending positive fp32 values shaped like a row of pairwise distances of the `scaled` family."""
    v = np.sort((1.0 + spread * rng.random(n)).astype(np.float32) * np.float32(rng.uniform(0.5, 2000.0)))
    return v.view(np.uint32).copy()


def main():
    rng = np.random.default_rng(7)
    for n, take, picks in ((3040, 3040, 300), (7600, 7600, 300)):
        vals = distance_like_row(rng, n + 200)
        rec = Record(vals, take)
        assert rec.s == rec.literal()
        costs, fallbacks = [], 0
        for pick in range(picks):
            live = np.flatnonzero(vals[:rec.end])
            if rng.random() < 0.85:     # the winner is usually close to this row: among its first few hundred entries
                k = int(live[min(len(live) - 1, int(rng.exponential(150.0)))])
                c = rec.mark(k)
            else:
                c = rec.mark(None)
            assert rec.s == rec.literal(), 'pick %d: incremental %08x, literal %08x' % (pick, rec.s, rec.literal())
            if c < 0:
                fallbacks += 1
            else:
                costs.append(c)
        print('prefix %d: %d updates exact; %d fell back to the full chain; table entries read per update: median %d, mean %.0f, '
              'max %d; events held at the end: %d crossings, %d ties'
              % (take, picks, fallbacks, np.median(costs), np.mean(costs), max(costs),
                 sum(ev[1] == 'cross' for ev in rec.events), sum(ev[1] == 'tie' for ev in rec.events)))


if __name__ == '__main__':
    main()
