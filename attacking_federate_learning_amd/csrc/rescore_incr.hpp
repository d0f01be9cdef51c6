// Exact INCREMENTAL update of the reference's Krum score between two picks of the Bulyan loop (defences.py:33-34, 59-68).
//
// The score is Python's sum() over np.float32 values in ascending order: a chain of round-to-nearest-even additions.  Between
// two picks a row's prefix changes by ONE entry -- the winner's distance is marked (it adds nothing from then on) or, if the
// winner lay behind the prefix, the prefix loses its last live entry -- and re-adding the whole chain for that is what the
// loop's re-score costs today (select.hip: reference_score_marked).  While the running sum s = I q stays inside one binade
// (q its ulp, 2^23 <= I < 2^24),
//     fl(s + x) = (I + a + t) q,   a = floor(x / q),   t = [rem > q/2] or [rem == q/2 and I + a odd]:
// the increment of an entry depends on the entry and the unit only, except at a TIE (the parity of I in front of it) and at a
// CROSSING into the next binade (an fp32 addition, unit 2q from there on).  So marking entry k lowers every later partial sum
// by one integer `shift` (k's own increment) until the next event; a tie changes an odd shift by +-1; around a crossing the
// two chains are in different binades for an entry or two and are added literally until they meet again.  A Record holds the
// chain's sum and its events behind a literal head; `mark` / `drop_last` update both, or say that they cannot (-1: the caller
// re-scores in full, which is always right).
//
// Scalar code, host and device: the GPU runs it wave-uniformly (one wave per contender, every lane the same values), the CPU
// tests (tests/test_rescore_incr_native.py) compile it with g++ and check it bit for bit against the literal chain; the
// Python model of the same algorithm is scripts/proto/seqsum_incr.py.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define BYZ_INCR_HD __host__ __device__ __forceinline__
#else
#define BYZ_INCR_HD inline
#endif

namespace byz {
namespace incr {

constexpr int kMaxEvents = 24;     // ties and crossings a record can hold behind its head (more: the record is not kept)
constexpr int kMeetLimit = 16;     // literal steps a crossing region may take
constexpr int kMaxFresh = 8;       // events a crossing region may leave
constexpr uint32_t kGone = 0x80000000u;   // -0.0f: a removed entry (select.hip marks the table with it)

enum Kind : int { kPlain = 0, kTie = 1, kCross = 2 };

struct Event {
    int32_t pos;         // physical position in the row's table of ascending values
    uint32_t kind_t;     // bit 31: crossing; bit 0: how a tie was resolved (its t)
    uint32_t before;     // crossing: the sum in front of the entry ...
    uint32_t after;      // ... and behind it
};

struct Record {
    uint32_t s;          // the score (bits)
    uint32_t s_head;     // the sum in front of physical position head_end
    int32_t head_end;    // [0, head_end) is always re-added literally; events are kept for positions >= head_end
    int32_t end;         // physical end of the prefix
    int32_t valid_pick;  // the pick `s` belongs to; -1: no record
    int32_t n_events;
    Event ev[kMaxEvents];
};

BYZ_INCR_HD bool adds_nothing(uint32_t xb) { return (xb & 0x7fffffffu) == 0u; }   // +0.0 (a live twin) or the -0.0 mark

BYZ_INCR_HD uint32_t fadd_bits(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(__fadd_rn(__uint_as_float(a), __uint_as_float(b)));
#else
    union { uint32_t u; float f; } x, y, z;
    x.u = a;
    y.u = b;
    volatile float sum = x.f + y.f;
    z.f = sum;
    return z.u;
#endif
}

// value = I * 2^eq with 2^23 <= I < 2^24 (normal) or I < 2^23, eq = -149 (subnormal / zero)
BYZ_INCR_HD void decompose(uint32_t bits, uint32_t& I, int& eq) {
    const uint32_t e = (bits >> 23) & 0xffu, frac = bits & 0x7fffffu;
    if (e == 0u) {
        I = frac;
        eq = -149;
    } else {
        I = frac | 0x800000u;
        eq = static_cast<int>(e) - 150;
    }
}

BYZ_INCR_HD uint32_t compose(uint32_t I, int eq) {
    if (eq == -149 && I < (1u << 23)) return I;
    return (static_cast<uint32_t>(eq + 150) << 23) | (I & 0x7fffffu);
}

// (a, above, tie) of one non-negative finite entry under the unit 2^eq; a saturates at 2^25 (a crossing for sure)
BYZ_INCR_HD void classify(uint32_t xb, int eq, uint32_t& a, uint32_t& above, uint32_t& tie) {
    uint32_t M;
    int E;
    decompose(xb, M, E);
    a = above = tie = 0u;
    if (M == 0u) return;
    const int sh = eq - E;
    if (sh <= 0) {
        a = sh <= -2 ? (1u << 25) : (M << (-sh));
        return;
    }
    if (sh >= 26) return;
    a = M >> sh;
    const uint32_t rem = M & ((1u << sh) - 1u), half = 1u << (sh - 1);
    above = rem > half ? 1u : 0u;
    tie = rem == half ? 1u : 0u;
}

// one addition of the chain and what kind of step it was (t: the rounding increment of a plain / tie step)
BYZ_INCR_HD uint32_t step(uint32_t s, uint32_t xb, int& kind, uint32_t& t) {
    kind = kPlain;
    t = 0u;
    if (adds_nothing(xb)) return s;
    const uint32_t nb = fadd_bits(s, xb);
    uint32_t I, a, above, tie, In;
    int eq, eqn;
    decompose(s, I, eq);
    classify(xb, eq, a, above, tie);
    t = above | (tie & ((I + a) & 1u));
    decompose(nb, In, eqn);
    if (I + a + t >= (1u << 24) || eqn != eq) {
        kind = kCross;
        t = 0u;
    } else {
        kind = tie ? kTie : kPlain;
    }
    return nb;
}

BYZ_INCR_HD bool is_cross(const Event& e) { return (e.kind_t >> 31) != 0u; }

BYZ_INCR_HD void erase_event(Record& r, int i) {
    for (int j = i; j + 1 < r.n_events; ++j) r.ev[j] = r.ev[j + 1];
    --r.n_events;
}

BYZ_INCR_HD bool insert_event(Record& r, int i, const Event& e) {
    if (r.n_events >= kMaxEvents) return false;
    for (int j = r.n_events; j > i; --j) r.ev[j] = r.ev[j - 1];
    r.ev[i] = e;
    ++r.n_events;
    return true;
}

// the unit of the running sum in front of position p >= head_end
BYZ_INCR_HD int unit_at(const Record& r, int p) {
    uint32_t I;
    int eq;
    decompose(r.s_head, I, eq);
    for (int i = 0; i < r.n_events && r.ev[i].pos < p; ++i)
        if (is_cross(r.ev[i])) decompose(r.ev[i].after, I, eq);
    return eq;
}

// The literal chain over vals(0 .. end), noting its events behind head_end: what a full re-score leaves behind.  Returns
// false when no record can be kept (a sign bit or a non-finite value inside the prefix, too many events); r.s is right
// either way.
template <class Vals>
BYZ_INCR_HD bool full(const Vals& vals, int end, int head_end, Record& r) {
    uint32_t s = 0u;
    bool ok = true;
    r.n_events = 0;
    r.head_end = head_end < end ? head_end : end;
    r.end = end;
    r.s_head = 0u;
    for (int p = 0; p < end; ++p) {
        if (p == r.head_end) r.s_head = s;
        const uint32_t xb = vals(p);
        if (adds_nothing(xb)) continue;
        if ((xb & 0x80000000u) != 0u || ((xb >> 23) & 0xffu) == 0xffu) ok = false;
        int kind;
        uint32_t t;
        const uint32_t nb = step(s, xb, kind, t);
        if (p >= r.head_end && kind != kPlain) {
            if (r.n_events < kMaxEvents) {
                Event e;
                e.pos = p;
                e.kind_t = kind == kCross ? 0x80000000u : t;
                e.before = s;
                e.after = nb;
                r.ev[r.n_events++] = e;
            } else {
                ok = false;
            }
        }
        s = nb;
    }
    if (end <= head_end) r.s_head = s;
    r.s = s;
    if (((s >> 23) & 0xffu) == 0xffu) ok = false;
    return ok;
}

// Entry k < r.end (value xk, already marked in the table) leaves the prefix.  s_head_new: the literal sum of [0, head_end)
// after the mark -- needed (and read) only when k < head_end.  0: r is the record of the new chain; -1: re-score in full.
template <class Vals>
BYZ_INCR_HD int mark(const Vals& vals, Record& r, int k, uint32_t xk, uint32_t s_head_new) {
    if (adds_nothing(xk)) return 0;   // a live +0.0 (a twin's distance): no partial sum moves
    uint32_t shift = 0u, s_new = 0u, s_old = 0u;
    int shift_eq = 0, at = 0, i = 0;
    bool literal = false;
    if (k < r.head_end) {
        s_old = r.s_head;
        s_new = s_head_new;
        r.s_head = s_head_new;
        if (r.end <= r.head_end) {
            r.s = s_new;
            return 0;
        }
        at = r.head_end;
        literal = true;
    } else {
        int hit = -1;
        for (int j = 0; j < r.n_events; ++j)
            if (r.ev[j].pos == k) hit = j;
        if (hit >= 0 && is_cross(r.ev[hit])) return -1;
        const int eq = unit_at(r, k);
        uint32_t a, above, tie;
        classify(xk, eq, a, above, tie);
        if (tie != 0u && hit < 0) return -1;   // (a tie the record does not know: it was not made from this table)
        const uint32_t t = hit >= 0 ? (r.ev[hit].kind_t & 1u) : above;
        if (hit >= 0) erase_event(r, hit);
        shift = a + t;
        shift_eq = eq;
        while (i < r.n_events && r.ev[i].pos <= k) ++i;
    }
    for (;;) {
        if (literal) {
            // both chains explicitly, entry by entry, until they are in one binade again and past the old chain's crossing
            Event fresh[kMaxFresh];
            int n_fresh = 0, steps = 0;
            for (;;) {
                uint32_t In, Io;
                int en, eo;
                decompose(s_new, In, en);
                decompose(s_old, Io, eo);
                const bool at_cross = i < r.n_events && r.ev[i].pos == at && is_cross(r.ev[i]);
                if (en == eo && !at_cross) break;
                if (at >= r.end) break;
                if (steps >= kMeetLimit) return -1;
                const uint32_t xb = vals(at);
                int kind, kind_old;
                uint32_t t, t_old;
                const uint32_t before = s_new;
                s_new = step(s_new, xb, kind, t);
                s_old = step(s_old, xb, kind_old, t_old);
                if (kind != kPlain) {
                    if (n_fresh >= kMaxFresh) return -1;
                    fresh[n_fresh].pos = at;
                    fresh[n_fresh].kind_t = kind == kCross ? 0x80000000u : t;
                    fresh[n_fresh].before = before;
                    fresh[n_fresh].after = s_new;
                    ++n_fresh;
                }
                ++steps;
                ++at;
                while (i < r.n_events && r.ev[i].pos < at) erase_event(r, i);   // the old chain's events inside the region
            }
            for (int j = 0; j < n_fresh; ++j)
                if (!insert_event(r, i + j, fresh[j])) return -1;
            i += n_fresh;
            if (at >= r.end) {
                r.s = s_new;
                return 0;
            }
            uint32_t In, Io;
            int en, eo;
            decompose(s_new, In, en);
            decompose(s_old, Io, eo);
            if (Io < In) return -1;   // (cannot happen: rounding is monotone and the new chain starts lower)
            shift = Io - In;
            shift_eq = en;
            literal = false;
        }
        if (i >= r.n_events) break;
        Event& ev = r.ev[i];
        if (!is_cross(ev)) {
            if ((shift & 1u) != 0u) {
                const uint32_t old_t = ev.kind_t & 1u, new_t = old_t ^ 1u;
                shift = shift + old_t - new_t;
                ev.kind_t = new_t;
                // (before / after of a tie are not used by the walk)
            }
            ++i;
            continue;
        }
        uint32_t Ib;
        int eb;
        decompose(ev.before, Ib, eb);
        if (eb != shift_eq || Ib < shift || Ib - shift < (1u << 23)) return -1;
        s_new = compose(Ib - shift, eb);
        s_old = ev.before;
        at = ev.pos;
        literal = true;
    }
    uint32_t If;
    int ef;
    decompose(r.s, If, ef);
    if (ef != shift_eq || If < shift) return -1;
    if (If - shift < (1u << 23) && ef != -149) return -1;
    r.s = compose(If - shift, ef);
    return 0;
}

// The winner lay behind the prefix: the prefix loses its last live entry.  0 / -1 as above.
template <class Vals>
BYZ_INCR_HD int drop_last(const Vals& vals, Record& r) {
    int p = r.end - 1, looked = 0;
    while (p >= 0 && vals(p) == kGone) {
        --p;
        if (++looked > kMeetLimit) return -1;
    }
    if (p < r.head_end) return -1;
    const uint32_t xb = vals(p);
    if (adds_nothing(xb)) {   // a live +0.0
        r.end = p;
        return 0;
    }
    if (r.n_events > 0 && r.ev[r.n_events - 1].pos == p) {
        const Event ev = r.ev[r.n_events - 1];
        --r.n_events;
        if (is_cross(ev)) {
            r.s = ev.before;
            r.end = p;
            return 0;
        }
        uint32_t I, a, above, tie;
        int eq;
        decompose(r.s, I, eq);
        classify(xb, eq, a, above, tie);
        const uint32_t inc = a + (ev.kind_t & 1u);
        if (I < inc || I - inc < (1u << 23)) return -1;
        r.s = compose(I - inc, eq);
        r.end = p;
        return 0;
    }
    uint32_t I, a, above, tie;
    int eq;
    decompose(r.s, I, eq);
    classify(xb, eq, a, above, tie);
    if (tie != 0u) return -1;
    const uint32_t inc = a + above;
    if (I < inc || I - inc < (1u << 23)) return -1;
    r.s = compose(I - inc, eq);
    r.end = p;
    return 0;
}

}  // namespace incr
}  // namespace byz
