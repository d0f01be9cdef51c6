"""numpy emulation of the Bulyan re-score's integer rule WITH ties (csrc/select.hip: reference_score).

The reference sums ascending fp32 values left to right (defences.py:33-34).  While the running sum s stays in one binade
[2^E, 2^(E+1)) it is S q with q = 2^(E-23) and S an integer in [2^23, 2^24), and fl(s + x) = q RNE(S + t), t = x / q (exact).
  * frac(t) != 1/2: the step adds rint(t), whatever S is;
  * frac(t) == 1/2: the step adds floor(t) + ((S + floor(t)) & 1) -- round half to even -- and leaves S EVEN.
So the parity of S in front of a tie is the XOR of the parities added since the previous tie of the chunk (0 after a tie)
or since the chunk's start (then it also carries the parity of the incoming S): a 64-entry chunk is an integer sum plus
one +-1 that depends on the incoming parity.  Only a binade crossing or an entry beyond 2^24 q needs the sequential chain.
This file replays that rule chunk by chunk exactly as the kernel does and compares it with the plain loop.
"""
import numpy as np


def sequential(values):
    s = np.float32(0.0)
    for v in values:
        s = np.float32(s + np.float32(v))
    return s


def chunk_stage_b(x, invq):
    """Per 64-entry chunk against the batch's binade: (big, tot0, delta)."""
    t = (x * invq).astype(np.float32)             # exact (power of two), may overflow to inf
    if not np.all(t < np.float32(16777216.0)):
        return True, 0, 0
    fl = np.floor(t)
    frac = t - fl
    tie = frac == np.float32(0.5)
    r = np.rint(t)                                  # ties to even on t alone: only used where frac != 1/2
    fli = fl.astype(np.int64)
    ri = r.astype(np.int64)
    c = np.where(tie, 0, ri & 1)                    # parity each non-tie entry adds
    contrib = np.where(tie, 0, ri).astype(np.int64)
    delta = 0
    last_tie = -1
    for l in np.nonzero(tie)[0]:
        seg = c[last_tie + 1:l]
        par = int(seg.sum() & 1)                    # parity in front of the tie, for an even incoming S (or after a tie)
        f = int(fli[l] & 1)
        contrib[l] = fli[l] + ((par ^ f) & 1)
        if last_tie < 0:
            delta = -1 if ((par ^ f) & 1) else 1    # an odd incoming S flips this one decision
        last_tie = l
    return False, int(contrib.sum()), delta


def by_rule(values, depth=8):
    values = np.asarray(values, dtype=np.float32)
    n = len(values)
    pad = (-n) % (64 * depth)
    xs = np.concatenate([values, np.zeros(pad, dtype=np.float32)]).reshape(-1, depth, 64)
    carry = np.float32(0.0)
    chains = fast = 0
    for batch in xs:
        k = 0
        while k < depth:
            eb = (carry.view(np.uint32) >> 23) & 0xff
            if not (40 <= eb <= 220):
                carry = chain(carry, batch[k])
                chains += 1
                k += 1
                continue
            invq = np.float32(2.0) ** np.float32(23 - (int(eb) - 127))
            q = np.float32(2.0) ** np.float32(int(eb) - 127 - 23)
            S = int(np.float32(carry * invq))
            assert (1 << 23) <= S < (1 << 24)
            stage = [chunk_stage_b(batch[j], invq) for j in range(depth)]
            while k < depth:
                big, tot0, delta = stage[k]
                if big:
                    carry = chain(carry, batch[k])
                    chains += 1
                    k += 1
                    break
                tot = tot0 + (delta if (S & 1) else 0)
                if S + tot < (1 << 24):
                    S += tot
                    carry = np.float32(np.float32(S) * q)
                    fast += 1
                    k += 1
                else:
                    carry = chain(carry, batch[k])
                    chains += 1
                    k += 1
                    break
    return carry, fast, chains


def chain(carry, chunk):
    s = np.float32(carry)
    for v in chunk:
        s = np.float32(s + v)
    return s


def main():
    rng = np.random.default_rng(0)
    bad = 0
    total_fast = total_chain = 0
    cases = 0
    for trial in range(400):
        n = int(rng.integers(1, 9000))
        kind = trial % 8
        if kind == 0:
            v = np.sqrt(rng.chisquare(16, n)).astype(np.float32)
        elif kind == 1:
            v = (rng.integers(1, 1 << 12, n) / 64.0).astype(np.float32)          # quantised: ties everywhere
        elif kind == 2:
            v = np.full(n, rng.uniform(0.1, 10.0), dtype=np.float32)             # one value: structural ties
        elif kind == 3:
            v = (rng.integers(0, 3, n) * 0.5 + rng.integers(0, 2, n) * 2.0 ** -12).astype(np.float32)
        elif kind == 4:
            v = (np.sqrt(rng.chisquare(16, n)) * 10.0 ** rng.integers(-20, 20)).astype(np.float32)
        elif kind == 5:
            v = np.concatenate([np.zeros(n // 3), rng.uniform(0, 1, n - n // 3)]).astype(np.float32)
        elif kind == 6:
            v = (rng.integers(1, 1 << 20, n).astype(np.float64) * 2.0 ** -9).astype(np.float32)
        else:
            v = np.exp(rng.uniform(-30, 30, n)).astype(np.float32)                # 26 decades: big entries, crossings
        v = np.sort(v)
        take = int(rng.integers(1, n + 1))
        v[take:] = 0.0                                                            # past the prefix: + 0.0
        want = sequential(v[:take])
        got, fast, chains = by_rule(v)
        cases += 1
        total_fast += fast
        total_chain += chains
        if want.view(np.uint32) != np.float32(got).view(np.uint32):
            bad += 1
            print('MISMATCH trial', trial, 'kind', kind, 'n', n, 'take', take, want, got)
    print('%d lists, %d mismatches; chunks by the integer rule %d, by the chain %d' % (cases, bad, total_fast, total_chain))


if __name__ == '__main__':
    main()
