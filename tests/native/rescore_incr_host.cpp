// Host harness of csrc/rescore_incr.hpp for tests/test_rescore_incr_native.py (compiled with g++ by the test; no GPU).
#include "rescore_incr.hpp"

using namespace byz::incr;

namespace {
struct Table {
    const uint32_t* v;
    int n;
    uint32_t operator()(int p) const { return v[p]; }
    uint32_t next_live(int pos, int& at) const {
        for (int p = pos + 1; p < n && p < pos + 16; ++p)
            if (!adds_nothing(v[p])) {
                at = p;
                return v[p];
            }
        return kNoValue;
    }
};
int g_fast_taken = 0, g_fast_declined = 0;
uint32_t head_sum(const uint32_t* vals, int head_end) {
    uint32_t s = 0u;
    for (int p = 0; p < head_end; ++p) {
        int kind;
        uint32_t t;
        s = step(s, vals[p], kind, t);
    }
    return s;
}
}  // namespace

extern "C" {

int incr_fast_taken() { return g_fast_taken; }
int incr_fast_declined() { return g_fast_declined; }

int incr_record_bytes() { return static_cast<int>(sizeof(Record)); }

uint32_t incr_literal(const uint32_t* vals, int end) { return head_sum(vals, end); }

int incr_full(const uint32_t* vals, int end, int head_end, Record* r) {
    const bool ok = full(Table{vals, 1 << 30}, end, head_end, *r);
    r->valid_pick = ok ? 0 : -1;
    return ok ? 1 : 0;
}

// marks entry k (k >= 0) or drops the last live entry (k < 0).  1: updated from the events; 0: fell back to the full chain
// (the record is rebuilt); the sum in r->s is the new chain's either way.
int incr_update(uint32_t* vals, Record* r, int k) {
    int rc = -1;
    if (k >= 0) {
        const uint32_t xk = vals[k];
        vals[k] = kGone;
        if (r->valid_pick >= 0) {
            const Record saved = *r;
            if (mark_fast(Table{vals, r->end}, *r, k, xk) == 1) {
                ++g_fast_taken;
                rc = 0;
            } else {
                ++g_fast_declined;
                *r = saved;
                const uint32_t s_head_new = k < r->head_end ? head_sum(vals, r->head_end) : 0u;
                rc = mark(Table{vals, 1 << 30}, *r, k, xk, s_head_new);
            }
        }
    } else if (r->valid_pick >= 0) {
        rc = drop_last(Table{vals, 1 << 30}, *r);
    }
    if (rc == 0) return 1;
    int end = r->end;
    if (k < 0) {   // the literal meaning of "drop the last live entry"
        int p = end - 1;
        while (p >= 0 && vals[p] == kGone) --p;
        end = p < 0 ? 0 : p;
    }
    const int head_end = r->head_end;
    const bool ok = full(Table{vals, 1 << 30}, end, head_end > 0 ? head_end : 512, *r);
    r->valid_pick = ok ? 0 : -1;
    return 0;
}

}  // extern "C"
