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

// (development: a translation unit may define BYZ_INCR_PROBE(i) to count / time the sections of an update)
#ifndef BYZ_INCR_PROBE
#define BYZ_INCR_PROBE(i) do { } while (0)
#endif

namespace byz {
namespace incr {

constexpr int kMaxEvents = 24;     // ties and crossings a record can hold behind its head (more: the record is not kept)
constexpr int kMeetLimit = 16;     // literal steps a crossing region may take
constexpr int kMaxFresh = 8;       // events a crossing region may leave
constexpr uint32_t kGone = 0x80000000u;   // -0.0f: a removed entry (select.hip marks the table with it)
constexpr uint32_t kNoValue = 0xffffffffu;   // what a table accessor returns for a position it cannot reach (the GPU's prefetched windows)

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

    // (the algorithms below reach the events through these five only: the GPU keeps event i in lane i of the wave instead)
    BYZ_INCR_HD Event get(int i) const { return ev[i]; }
    BYZ_INCR_HD void set(int i, const Event& e) { ev[i] = e; }
    BYZ_INCR_HD void erase(int i) {
        for (int j = i; j + 1 < n_events; ++j) ev[j] = ev[j + 1];
        --n_events;
    }
    BYZ_INCR_HD bool insert(int i, const Event& e) {
        if (n_events >= kMaxEvents) return false;
        for (int j = n_events; j > i; --j) ev[j] = ev[j - 1];
        ev[i] = e;
        ++n_events;
        return true;
    }
    BYZ_INCR_HD int find(int pos) const {   // the event at physical position pos, or -1
        for (int j = 0; j < n_events; ++j)
            if (ev[j].pos == pos) return j;
        return -1;
    }
    BYZ_INCR_HD int first_after(int pos) const {   // the first event behind physical position pos (n_events: none)
        int j = 0;
        while (j < n_events && ev[j].pos <= pos) ++j;
        return j;
    }
    BYZ_INCR_HD int last_cross_before(int pos) const {   // the last crossing in front of physical position pos, or -1
        int hit = -1;
        for (int j = 0; j < n_events && ev[j].pos < pos; ++j)
            if ((ev[j].kind_t >> 31) != 0u) hit = j;
        return hit;
    }
    BYZ_INCR_HD int next_relevant(int i, bool ties_too) const {   // the first event >= i that is a crossing (or, ties_too, any)
        while (i < n_events && !ties_too && (ev[i].kind_t >> 31) == 0u) ++i;
        return i;
    }
};

BYZ_INCR_HD bool adds_nothing(uint32_t xb) { return (xb & 0x7fffffffu) == 0u; }   // +0.0 (a live twin) or the -0.0 mark

BYZ_INCR_HD uint32_t fadd_bits(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // (wave-uniform in, wave-uniform out: saying so keeps the integer arithmetic around it on the scalar unit)
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(__fadd_rn(__uint_as_float(a), __uint_as_float(b))))));
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

// the unit of the running sum in front of position p >= head_end
template <class Rec>
BYZ_INCR_HD int unit_at(const Rec& r, int p) {
    uint32_t I;
    int eq;
    const int c = r.last_cross_before(p);
    decompose(c >= 0 ? r.get(c).after : r.s_head, I, eq);
    return eq;
}

// The literal chain over vals(0 .. end), noting its events behind head_end: what a full re-score leaves behind.  Returns
// false when no record can be kept (a sign bit or a non-finite value inside the prefix, too many events); r.s is right
// either way.
template <class Vals, class Rec>
BYZ_INCR_HD bool full(const Vals& vals, int end, int head_end, Rec& r) {
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
            Event e;
            e.pos = p;
            e.kind_t = kind == kCross ? 0x80000000u : t;
            e.before = s;
            e.after = nb;
            if (!r.insert(r.n_events, e)) ok = false;
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
template <class Vals, class Rec>
BYZ_INCR_HD int mark(const Vals& vals, Rec& r, int k, uint32_t xk, uint32_t s_head_new) {
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
        const int hit = r.find(k);
        Event at_k;
        at_k.kind_t = 0u;
        if (hit >= 0) at_k = r.get(hit);
        if (hit >= 0 && is_cross(at_k)) return -1;
        const int eq = unit_at(r, k);
        uint32_t a, above, tie;
        classify(xk, eq, a, above, tie);
        if (tie != 0u && hit < 0) return -1;   // (a tie the record does not know: it was not made from this table)
        const uint32_t t = hit >= 0 ? (at_k.kind_t & 1u) : above;
        if (hit >= 0) r.erase(hit);
        shift = a + t;
        shift_eq = eq;
        i = r.first_after(k);
    }
    BYZ_INCR_PROBE(0);   // prologue done
    for (;;) {
        if (literal) {
            // both chains explicitly, entry by entry, until they are in one binade again and past the old chain's crossing;
            // what the new chain does on the way goes on the record at once, the old chain's events on the way come off it
            int n_fresh = 0, steps = 0;
            for (;;) {
                uint32_t In, Io;
                int en, eo;
                decompose(s_new, In, en);
                decompose(s_old, Io, eo);
                bool at_cross = false;
                if (i < r.n_events) {
                    const Event e = r.get(i);
                    at_cross = e.pos == at && is_cross(e);
                }
                if (en == eo && !at_cross) break;
                if (at >= r.end) break;
                if (steps >= kMeetLimit) return -1;
                BYZ_INCR_PROBE(1);   // a literal iteration
                const uint32_t xb = vals(at);
                if (xb == kNoValue) return -1;
                int kind;
                uint32_t t;
                const uint32_t before = s_new;
                s_new = step(s_new, xb, kind, t);
                if (!adds_nothing(xb)) s_old = fadd_bits(s_old, xb);   // (of the old chain only the value is needed)
                ++steps;
                ++at;
                while (i < r.n_events && r.get(i).pos < at) r.erase(i);   // the old chain's events up to this entry
                if (kind != kPlain) {
                    Event e;
                    e.pos = at - 1;
                    e.kind_t = kind == kCross ? 0x80000000u : t;
                    e.before = before;
                    e.after = s_new;
                    if (++n_fresh > kMaxFresh || !r.insert(i, e)) return -1;
                    ++i;
                }
            }
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
        i = r.next_relevant(i, (shift & 1u) != 0u);   // (an even shift leaves every tie as it was resolved)
        if (i >= r.n_events) break;
        Event ev = r.get(i);
        if (!is_cross(ev)) {
            if ((shift & 1u) != 0u) {
                const uint32_t old_t = ev.kind_t & 1u, new_t = old_t ^ 1u;
                shift = shift + old_t - new_t;
                ev.kind_t = new_t;
                r.set(i, ev);   // (before / after of a tie are not used by the walk)
            }
            ++i;
            continue;
        }
        uint32_t Ib;
        int eb;
        decompose(ev.before, Ib, eb);
        if (eb != shift_eq || Ib < shift || Ib - shift < (1u << 23)) return -1;
        s_new = compose(Ib - shift, eb);
        {
            // the usual case: the new chain crosses at the very entry the old one did -- the two meet right behind it, the
            // event stays where it is with the new chain's sums, and the shift is read off in the new unit
            const uint32_t xb = vals(ev.pos);
            if (xb == kNoValue) return -1;
            const uint32_t after_new = fadd_bits(s_new, xb);
            uint32_t In, Io;
            int en, eo;
            decompose(after_new, In, en);
            decompose(ev.after, Io, eo);
            if (en == eo && Io >= In) {
                BYZ_INCR_PROBE(2);   // a crossing met at its own entry
                ev.before = s_new;
                ev.after = after_new;
                r.set(i, ev);
                shift = Io - In;
                shift_eq = en;
                ++i;
                continue;
            }
        }
        {
            // the other usual case: the new chain, one entry's worth lower, takes the old crossing entry without leaving the
            // binade and crosses at the NEXT entry -- where the two meet.  (Anything else -- a tie on the way, an entry that adds
            // nothing, another event of the old chain right there -- is the literal region's business.)
            const uint32_t x0 = vals(ev.pos), x1 = ev.pos + 1 < r.end ? vals(ev.pos + 1) : kNoValue;
            bool next_is_event = false;
            if (i + 1 < r.n_events) next_is_event = r.get(i + 1).pos == ev.pos + 1;
            if (x0 != kNoValue && x1 != kNoValue && !adds_nothing(x1) && !next_is_event) {
                uint32_t a0, above0, tie0, I1, I2, Io;
                int e1, e2, eo;
                classify(x0, eb, a0, above0, tie0);
                const uint32_t mid = fadd_bits(s_new, x0);
                decompose(mid, I1, e1);
                if (tie0 == 0u && e1 == eb) {
                    const uint32_t after_new = fadd_bits(mid, x1), after_old = fadd_bits(ev.after, x1);
                    decompose(after_new, I2, e2);
                    decompose(after_old, Io, eo);
                    uint32_t Ia;
                    int ea;
                    decompose(ev.after, Ia, ea);
                    if (e2 == eo && eo == ea && e2 != eb && Io >= I2) {
                        BYZ_INCR_PROBE(5);   // a crossing that moved to the next entry
                        ev.pos = ev.pos + 1;
                        ev.before = mid;
                        ev.after = after_new;
                        r.set(i, ev);
                        shift = Io - I2;
                        shift_eq = e2;
                        ++i;
                        continue;
                    }
                }
            }
        }
        BYZ_INCR_PROBE(3);   // a crossing that needs the literal region
        s_old = ev.before;
        at = ev.pos;
        literal = true;
    }
    BYZ_INCR_PROBE(4);   // the walk is over
    uint32_t If;
    int ef;
    decompose(r.s, If, ef);
    if (ef != shift_eq || If < shift) return -1;
    if (If - shift < (1u << 23) && ef != -149) return -1;
    r.s = compose(If - shift, ef);
    return 0;
}

// The same update for the case that nearly every pick is: the marked entry lies behind the literal head, is no
// event and no tie itself; every later crossing is met at its own entry or at the next live one, with no tie and no other event in
// between.  Straight-line work per event -- the sums inside one binade are moved as integers on their bit patterns.  1: r is the
// record of the new chain; 0: not this case -- r may be half-way changed, the caller restores its copy and runs `mark`.
// vals.next_live(pos, at): the first entry behind pos that adds something (its position to `at`), kNoValue if out of reach.
template <class Vals, class Rec>
BYZ_INCR_HD int mark_fast(const Vals& vals, Rec& r, int k, uint32_t xk) {
    if (adds_nothing(xk)) return 1;
    if (k < r.head_end || r.find(k) >= 0) return 0;
    uint32_t a, above, tie;
    classify(xk, unit_at(r, k), a, above, tie);
    if (tie != 0u) return 0;
    uint32_t shift = a + above;
    int i = r.first_after(k);
    for (;;) {
        i = r.next_relevant(i, (shift & 1u) != 0u);
        if (i >= r.n_events) break;
        Event ev = r.get(i);
        if (!is_cross(ev)) {   // (only an odd shift gets here)
            const uint32_t old_t = ev.kind_t & 1u, new_t = old_t ^ 1u;
            shift = shift + old_t - new_t;
            ev.kind_t = new_t;
            r.set(i, ev);
            ++i;
            continue;
        }
        const uint32_t eb = ev.before >> 23, ea = ev.after >> 23;
        if ((ev.before & 0x7fffffu) < shift && eb != 0u) return 0;   // the new chain would start this stretch a binade lower
        if (eb == 0u) return 0;                                      // (subnormal sums: the general walk)
        const uint32_t s_new = ev.before - shift;
        const uint32_t x0 = vals(ev.pos);
        if (x0 == kNoValue) return 0;
        const uint32_t after0 = fadd_bits(s_new, x0);
        if ((after0 >> 23) == ea) {   // crossed at the same entry: the two chains meet right behind it
            if (ev.after < after0) return 0;
            shift = ev.after - after0;
            ev.before = s_new;
            ev.after = after0;
            r.set(i, ev);
            ++i;
            continue;
        }
        if ((after0 >> 23) != eb) return 0;
        uint32_t a0, above0, tie0;
        classify(x0, static_cast<int>(eb) - 150, a0, above0, tie0);
        if (tie0 != 0u) return 0;   // (the entry is a plain step of the new chain now; a tie would have to go on the record)
        int at1 = 0;
        const uint32_t x1 = vals.next_live(ev.pos, at1);
        if (x1 == kNoValue || at1 >= r.end) return 0;
        if (i + 1 < r.n_events && r.get(i + 1).pos <= at1) return 0;
        const uint32_t after_new = fadd_bits(after0, x1), after_old = fadd_bits(ev.after, x1);
        if ((after_new >> 23) != ea || (after_old >> 23) != ea || after_old < after_new) return 0;
        shift = after_old - after_new;
        ev.pos = at1;
        ev.before = after0;
        ev.after = after_new;
        r.set(i, ev);
        ++i;
    }
    if ((r.s >> 23) == 0u || (r.s & 0x7fffffu) < shift) return 0;
    // (the stretch behind the last crossing is the final binade: its unit is the shift's)
    r.s = r.s - shift;
    return 1;
}

// The winner lay behind the prefix: the prefix loses its last live entry.  0 / -1 as above.
template <class Vals, class Rec>
BYZ_INCR_HD int drop_last(const Vals& vals, Rec& r) {
    int p = r.end - 1, looked = 0;
    while (p >= 0 && vals(p) == kGone) {
        --p;
        if (++looked >= kMeetLimit) return -1;
    }
    if (p < r.head_end) return -1;
    const uint32_t xb = vals(p);
    if (xb == kNoValue) return -1;
    if (adds_nothing(xb)) {   // a live +0.0
        r.end = p;
        return 0;
    }
    Event ev;
    ev.pos = -1;
    if (r.n_events > 0) ev = r.get(r.n_events - 1);
    if (ev.pos == p) {
        r.erase(r.n_events - 1);
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
