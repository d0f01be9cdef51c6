// Median-window trimmed mean by counting selection, reference defences.py:44-52, per parameter (column):
//
//   med  = np.median(column)                       fp32; even count -> (a + b) / 2
//   good = sorted(column - med, key=abs)[:k]       stable: ties in |x - med| keep the lower row first
//   out  = np.mean(good) + med
//
// The reference sorts every column; nothing in the result needs the full order.  Only three order
// statistics matter: the median (one or two middle ranks) and t = the k-th smallest |x - med|.  Given t,
//     sum(good) = sum of d over {|d| <= t}                          when exactly k values satisfy |d| <= t,
// and otherwise (ties at t beyond the k-th) the tied values are taken in row order, which only matters when
// +t and -t both occur.
//
// An order statistic of R values held in registers is found by counting probes: count(T) = #{x <= T}.
//   * One probe costs 2.5 VALU operations per value: d = T - x, the sign bit of d, a three-input add.  On
//     gfx950 a VOPC compare issues at half rate and the ballot/s_bcnt1/s_add form is bound by the scalar
//     unit (scripts/ubench/count_rate.hip: 10.4 cycles per 64 values, against 6.3 for the arithmetic form).
//   * Probes are chosen by safeguarded interpolation on the counts (the bracket ends carry their counts, so
//     the empirical CDF is known at both), aiming alternately just above and just below the wanted rank so
//     that two probes sandwich it; when a probe fails to shrink the candidate set by 15%, the next threshold
//     is a data value from inside the bracket (a quickselect step: it splits by rank, so outliers of 1e30
//     or heavy tails cannot stall the search); after 12 arithmetic probes the midpoint of the
//     order-preserving integer keys takes over, which bounds the worst case at 12 + 32 probes.
//     scripts/proto/probe_policy.py: 4.4 + 8.8 probes for N(0,1) at n = 1000, k = 799 (plain bisection
//     7.9 + 6.8); 4.7 + 3.5 at n = 2080, k = 159 (9.1 + 9.1); 7.6 + 9.5 with +-1e30 outliers (20 + 25).
//   * When at most 16 candidates remain inside the bracket they are compacted into the column's 16-lane
//     segment of one register; one 16-lane bitonic network (DPP only) then sorts the candidates of the
//     wave's four columns at once.
//
// Data movement (gfx950): one workgroup = 8 waves = one tile of 32 consecutive columns x all rows.
//   * global loads are 128-byte row segments (8 lanes x dwordx4), 4 in flight per thread, software
//     pipelined against the LDS transit of the previous 256-row chunk;
//   * the LDS transit buffer (256 rows x 36 floats, 36 KB) only transposes: wave w takes the 4 columns
//     4w..4w+3, lane l takes rows l, l+64, ... with conflict-free ds_read_b128, so a lane ends up with
//     RPL = ceil(R/64) values of each of its 4 columns in registers.  The tile lives in the register file
//     (512 KB per CU), which is what lets R reach 2560 rows; LDS (160 KB) could not hold it;
//   * after staging, a wave never synchronises with another wave again: its 4 columns are entirely its own.
// Rows past R are padded with +inf: T - inf has its sign bit set for every finite probe, so the padding is
// never counted, and |inf - med| is never inside a window.  A NaN in the data makes np.median, and with it
// the reference's result, NaN: a cheap per-thread x*0 accumulation flags non-finite input, and only a
// flagged tile pays for the exact per-column NaN test.
//
// Algorithmic traffic: 4 bytes read per (row, column), 4 bytes written per column.  Bound: HBM.
#include "common.hpp"

#include "lane_exchange.hpp"

#include <cstdlib>
#include <cstring>

namespace byz {
namespace {

using namespace lanes;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCols = 16;                 // columns per tile (4 waves x 4 columns, or 16 waves x 1 column)
constexpr int kStride = 20;               // floats per LDS row: 16-byte aligned, b128 column reads conflict-free
constexpr int kMaxRpl = 40;               // 2560 rows
constexpr int kArithProbes = 12;          // value-space probes before switching to key-space midpoints
constexpr int kCand = 16;                 // candidates per column handed to the sorting network
constexpr float kAimOffset = 0.35f * kCand;   // ranks by which a probe aims past the wanted rank
constexpr float kStallRatio = 0.85f;

__device__ __forceinline__ uint32_t fkey(float v) {  // order-preserving map float -> uint32
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_fkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

template <bool ABS>
__device__ __forceinline__ float mag(float v) { return ABS ? __builtin_fabsf(v) : v; }

// wave-uniform values are moved to the scalar file explicitly: the probe control flow is scalar code
__device__ __forceinline__ float uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, m, 64));
    return uniform(v);
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = __builtin_fminf(v, __shfl_xor(v, m, 64));
    return uniform(v);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return uniform(v);
}
// integer wave sum: four DPP butterfly steps inside each 16-lane row, then four v_readlane
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    int x = static_cast<int>(v);
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);   // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) +
                                 __builtin_amdgcn_readlane(x, 32) + __builtin_amdgcn_readlane(x, 48));
}

// ascending sort inside every 16-lane segment (10 compare-exchange stages; lane masks <= 15)
__device__ __forceinline__ float segment16_sort(float v, int lane) {
    const float pinf = __builtin_inff();
    for_pow2_up<2, 16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        {
            const float sel = (lane & (k / 2)) ? pinf : -pinf;   // the upper partner keeps the maximum
            v = __builtin_amdgcn_fmed3f(v, lane_xor(v, k - 1, lane), sel);
        }
        for_pow2_down<k / 4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float sel = (lane & j) ? pinf : -pinf;
            v = __builtin_amdgcn_fmed3f(v, lane_xor(v, j, lane), sel);
        });
    });
    return v;
}

// Bracket of one selection for one column; the fields the caller fills in and reads back are wave-uniform.
struct Bracket {
    float lo, hi;      // count(lo) = c_lo <= r < c_hi = count(hi)   (count(T) = #{mag(x) <= T})
    int c_lo, c_hi;
    float a, b;        // results: rank r and rank r + 1
};

template <typename T>
__device__ __forceinline__ T by_column(int col, T v0, T v1, T v2, T v3) {
    return col == 0 ? v0 : (col == 1 ? v1 : (col == 2 ? v2 : v3));
}
__device__ __forceinline__ float lane_value(float v, int src) {   // src is wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Ranks r (and r + 1 with want_next) of mag(x[c]) for the wave's four columns at once.
// On entry q[c].lo / hi / c_lo / c_hi describe a valid bracket.  slots = 64 * registers in use.
//
// The probe bookkeeping of the four columns runs as ONE vector instruction stream: lane l carries the
// bracket of column l & 3, so choosing the four next probes and absorbing the four counts costs one pass
// of ~100 vector instructions, not four scalar ones with their branches.  Only the counting passes, the
// compaction and the rare pivot search are per column.
template <int RPL, int NC, bool ABS>
__device__ __forceinline__ void select4(const float (&x)[NC][RPL], int groups, int slots, int r, bool want_next,
                                        Bracket (&q)[4], int lane, float* strip) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;
    const int col = lane & 3;
    const float pinf = __builtin_inff();
    const int r_hi = want_next ? r + 1 : r;
    const float want = static_cast<float>(r) + (want_next ? 1.0f : 0.5f);
    // per-lane state of column `col`
    float lo = by_column(col, q[0].lo, q[1].lo, q[2].lo, q[3].lo);
    float hi = by_column(col, q[0].hi, q[1].hi, q[2].hi, q[3].hi);
    int c_lo = by_column(col, q[0].c_lo, q[1].c_lo, q[2].c_lo, q[3].c_lo);
    int c_hi = by_column(col, q[0].c_hi, q[1].c_hi, q[2].c_hi, q[3].c_hi);
    int state = col >= NC ? 2 : ((c_hi - c_lo <= kCand) ? 1 : 0);   // 0 probing, 1 <= kCand candidates, 2 resolved, 3 split at T
    int last = 0, stalled = 0, probes = 0;
    float T = hi, res_a = 0.0f, res_b = 0.0f;

    for (int guard = 0; guard < 64; ++guard) {
        if ((__ballot(state == 0) & 0xFull) == 0ull) break;
        // ---- next probe of every probing column
        {
            const bool active = state == 0;
            const uint32_t klo = fkey(lo), khi = fkey(hi);
            const bool adjacent = khi - klo <= 1u;       // no float strictly inside: every candidate equals hi
            const float cand = static_cast<float>(c_hi - c_lo);
            const float aim = want + (last > 0 ? -kAimOffset : kAimOffset);
            float f = (aim - static_cast<float>(c_lo)) * __builtin_amdgcn_rcpf(cand);
            f = __builtin_fminf(__builtin_fmaxf(f, 0.02f), 0.98f);
            f = stalled ? 0.5f : f;
            const float Ta = __builtin_fmaf(f, hi - lo, lo);
            const bool arith = probes < kArithProbes && finite_f(lo) && finite_f(hi) && Ta > lo && Ta < hi;
            const float Tk = from_fkey(klo + ((khi - klo) >> 1));
            if (active) T = (arith ? Ta : Tk) + 0.0f;   // -0.0 would count +0.0 as greater; a split column keeps its T
            if (active && adjacent) {
                state = 2;
                res_a = hi;
                res_b = hi;
            }
            // quickselect step for a stalled column: a data value from inside the bracket, looked for in the
            // column's first registers only (rare path, per column)
            const unsigned pivots = static_cast<unsigned>(__ballot(state == 0 && stalled && arith) & 0xFull);
            if (pivots) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if ((pivots >> c) & 1u) {
                        const float lo_c = lane_value(lo, c), hi_c = lane_value(hi, c);
                        bool have = false;
                        float pv = 0.0f;
#pragma unroll
                        for (int jj = 0; jj < GS; ++jj) {
                            if (!have) {
                                const float a = mag<ABS>(x[c][jj]);
                                const unsigned long long m = __ballot(a > lo_c && a <= hi_c);
                                if (m) {
                                    pv = lane_value(a, __builtin_ctzll(m));
                                    if (!(pv < hi_c)) pv = from_fkey(fkey(pv) - 1u);   // stay strictly inside
                                    have = pv > lo_c;
                                }
                            }
                        }
                        if (have && col == c) T = pv + 0.0f;
                    }
                }
            }
            probes += (state == 0) ? 1 : 0;
        }
        // ---- counting pass: for each column still probing, the values greater than its probe
        uint32_t neg[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (__builtin_amdgcn_readlane(state, c) == 0) {
                const float Tc = lane_value(T, c);
                uint32_t n0 = 0u;
#pragma unroll
                for (int g = 0; g < RPL / GS; ++g) {
                    if (g < groups) {
#pragma unroll
                        for (int jj = 0; jj < GS; ++jj)
                            n0 += __float_as_uint(Tc - mag<ABS>(x[c][g * GS + jj])) >> 31;
                    }
                }
                neg[c] = n0;
            }
        }
        // per-lane counts are at most 40: two columns share one register through the reduction
        const uint32_t s01 = wave_sum_u32(neg[0] | (neg[1] << 16));
        const uint32_t s23 = wave_sum_u32(neg[2] | (neg[3] << 16));
        // ---- absorb the counts
        {
            const uint32_t packed = (lane & 2) ? s23 : s01;
            const int greater = static_cast<int>((lane & 1) ? (packed >> 16) : (packed & 0xffffu));
            const int c = slots - greater;
            const bool active = state == 0;
            const int before = c_hi - c_lo;
            const bool to_lo = active && c <= r;
            const bool to_hi = active && c > r_hi;
            lo = to_lo ? T : lo;
            c_lo = to_lo ? c : c_lo;
            hi = to_hi ? T : hi;
            c_hi = to_hi ? c : c_hi;
            last = to_lo ? -1 : (to_hi ? 1 : last);
            const int after = c_hi - c_lo;
            stalled = static_cast<float>(after) > kStallRatio * static_cast<float>(before) ? 1 : 0;
            if (active) state = (!to_lo && !to_hi) ? 3 : (after <= kCand ? 1 : 0);   // 3: T separates ranks r, r + 1
        }
    }
    // ---- a probe that fell exactly between ranks r and r + 1: the neighbours on both sides
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (__builtin_amdgcn_readlane(state, c) == 3) {
            const float Tc = lane_value(T, c);
            float below = -pinf, above = pinf;
#pragma unroll
            for (int g = 0; g < RPL / GS; ++g) {
                if (g < groups) {
#pragma unroll
                    for (int jj = 0; jj < GS; ++jj) {
                        const float a = mag<ABS>(x[c][g * GS + jj]);
                        below = (a <= Tc) ? __builtin_fmaxf(below, a) : below;
                        above = (a > Tc) ? __builtin_fminf(above, a) : above;   // +inf padding loses every min
                    }
                }
            }
            const float wa = wave_max(below), wb = wave_min(above);
            if (col == c) {
                res_a = wa;
                res_b = wb;
                state = 2;
            }
        }
    }
    // ---- compaction: the candidates of column c go to lanes 16 c .. 16 c + 15 of the strip
    int valid[4] = {0, 0, 0, 0};
    const unsigned to_sort = static_cast<unsigned>(__ballot(state == 1) & 0xFull);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if ((to_sort >> c) & 1u) {
            const float lo_next = uniform(from_fkey(fkey(lane_value(lo, c)) + 1u)), hi_c = lane_value(hi, c);
            int n_cand = 0;
#pragma unroll
            for (int g = 0; g < RPL / GS; ++g) {
                if (g < groups) {
#pragma unroll
                    for (int jj = 0; jj < GS; ++jj) {
                        const float a = mag<ABS>(x[c][g * GS + jj]);
                        const bool in = __builtin_amdgcn_fmed3f(a, lo_next, hi_c) == a;   // lo < a <= hi
                        const unsigned long long m = __ballot(in);
                        if (m) {  // uniform
                            const int pos = n_cand + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                                __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                            if (in && pos < kCand) strip[c * kCand + pos] = a;
                            n_cand += __popcll(m);
                        }
                    }
                }
            }
            valid[c] = n_cand < kCand ? n_cand : kCand;
        }
    }
    if (to_sort) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int seg = lane >> 4;
        const int n_valid = by_column(seg, valid[0], valid[1], valid[2], valid[3]);
        float s = (lane & 15) < n_valid ? strip[lane] : pinf;
        s = segment16_sort(s, lane);
        __builtin_amdgcn_wave_barrier();   // the strip is reused by the next selection
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if ((to_sort >> c) & 1u) {
                const int i = c * kCand + (r - __builtin_amdgcn_readlane(c_lo, c));
                const float sa = lane_value(s, i & 63), sb = lane_value(s, (i + 1) & 63);
                if (col == c) {
                    res_a = sa;
                    res_b = want_next ? sb : sa;
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        q[c].a = lane_value(res_a, c);
        q[c].b = lane_value(res_b, c);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Bucket selection: the fast path of both selections.
//
// One pass puts every value of a column into one of 64 equal-width buckets of [lo, hi] (an LDS histogram per wave and
// column, ds_add_u32), a 64-lane prefix sum finds the bucket that holds the wanted rank, and if that bucket has at most
// 64 values they are compacted and sorted by one 64-lane bitonic network; otherwise a second level splits that one
// bucket into 64 again.  Gaussian columns of 1000 values need one level (the busiest bucket holds ~45 values), 2560
// values need two.  That is ~13 vector instructions per value and selection against ~45 for the probing search below,
// which stays as the general path: any column the buckets cannot resolve (outliers that squeeze everything into one
// bucket, non-finite ranges, more than 64 equal values) sends the wave's columns through select4.
//
// The bucket index floor((a - lo) * inv) is monotone in a, so "all values in lower buckets" are exactly the values that
// sort before the target bucket: the rank bookkeeping is exact whatever the rounding of the index arithmetic does.
// Padding (+inf) lands in an extra bucket 64 that nothing reads.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kBuckets = 64;
// The target bucket must fit one 64-lane sort.  With rows/64 values per bucket on average the central bucket of a
// bell-shaped column overflows from about 1500 rows, and a failed attempt is pure overhead (measured at theta = 2080:
// 74 ms without the attempt, 110 ms with it), so the fast path is compiled only for tiles of at most this many rows
// per lane.
#ifndef BYZ_BUCKET_MAX_RPL
#define BYZ_BUCKET_MAX_RPL 16
#endif
constexpr int kBucketMaxRpl = BYZ_BUCKET_MAX_RPL;
constexpr int kWaveScratch = 64 + (kBuckets + 1) * 4 + 4 * 64;   // floats per wave: strip, histogram, candidates

__device__ __forceinline__ uint32_t bucket_of(float a, float lo, float inv) {
    const float t = __builtin_fminf((a - lo) * inv, static_cast<float>(kBuckets));   // +inf padding -> bucket 64
    return static_cast<uint32_t>(t);   // v_cvt_u32_f32: negative values and NaN give 0
}

// For columns c with want[c]: ranks r (and r + 1 with want_next) of mag(x[c]); ok[c] says whether the column was resolved.
template <int RPL, int NC, bool ABS>
__device__ __forceinline__ void bucket_select(const float (&x)[NC][RPL], int groups, int r, bool want_next,
                                              const float (&lo0)[NC], const float (&hi0)[NC], const bool (&want)[NC],
                                              float (&res_a)[NC], float (&res_b)[NC], bool (&ok)[NC], int lane,
                                              float* scratch) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;
    // [column][bucket 0..64]: consecutive buckets are consecutive banks, so only lanes that hit the SAME bucket
    // serialise ([bucket][column] put every add of one instruction on 8 banks: SQ_LDS_BANK_CONFLICT = 1.7x the busy cycles)
    uint32_t* hist = reinterpret_cast<uint32_t*>(scratch + 64);
    constexpr int HS = kBuckets + 1;   // histogram stride per column
    float* cand = scratch + 64 + (kBuckets + 1) * 4;                        // [column][64]
    const float pinf = __builtin_inff();
    float lo[NC], inv[NC];          // current level's bucket map
    float plo[NC], pinv[NC];        // parent level's map and target bucket (level 2 only)
    uint32_t pb[NC];
    int below[NC];                  // values known to sort before the current range
    bool live[NC];                  // still being resolved by buckets
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        ok[c] = false;
        below[c] = 0;
        plo[c] = 0.0f; pinv[c] = 0.0f; pb[c] = 0u;
        const float width = hi0[c] - lo0[c];
        live[c] = want[c] && finite_f(lo0[c]) && finite_f(hi0[c]) && finite_f(width) && width > 0.0f;
        if (want[c] && finite_f(lo0[c]) && hi0[c] == lo0[c]) {   // every value is the same
            ok[c] = true;
            res_a[c] = res_b[c] = lo0[c];
        }
        lo[c] = lo0[c];
        inv[c] = live[c] ? uniform(static_cast<float>(kBuckets) * (1.0f - 1.0f / 1048576.0f) / width) : 0.0f;
        live[c] = live[c] && finite_f(inv[c]);
    }
#pragma unroll
    for (int level = 0; level < 2; ++level) {
        bool any = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) any = any || live[c];
        if (!any) break;
        // ---- histogram
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 4; ++c) hist[c * HS + lane] = 0u;
        if (lane < 4) hist[lane * HS + kBuckets] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (live[c]) {
#pragma unroll
                for (int g = 0; g < RPL / GS; ++g) {
                    if (g < groups) {
#pragma unroll
                        for (int jj = 0; jj < GS; ++jj) {
                            const float a = mag<ABS>(x[c][g * GS + jj]);
                            const uint32_t b = bucket_of(a, lo[c], inv[c]);
                            if (level == 0) {
                                atomicAdd(hist + c * HS + b, 1u);
                            } else if (bucket_of(a, plo[c], pinv[c]) == pb[c]) {   // only the parent bucket's values
                                atomicAdd(hist + c * HS + (b < kBuckets ? b : kBuckets - 1), 1u);
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u32x4 cnt4 = {hist[lane], hist[HS + lane], hist[2 * HS + lane], hist[3 * HS + lane]};   // lane = bucket
        // ---- inclusive prefix sums over the 64 buckets (two columns per register: counts stay below 65536)
        uint32_t p01 = cnt4[0] | (cnt4[1] << 16), p23 = cnt4[2] | (cnt4[3] << 16);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t u01 = __shfl_up(p01, d, 64), u23 = __shfl_up(p23, d, 64);
            if (lane >= d) {
                p01 += u01;
                p23 += u23;
            }
        }
        const uint32_t incl[4] = {p01 & 0xffffu, p01 >> 16, p23 & 0xffffu, p23 >> 16};
        // ---- target bucket(s) of every live column
        uint32_t tb[NC], tb1[NC];
        int n_in[NC];
        bool extract[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            extract[c] = false;
            tb[c] = tb1[c] = 0u;
            n_in[c] = 0;
            if (live[c]) {
                const int rr = r - below[c];                      // rank inside the current range
                const uint32_t excl = incl[c] - cnt4[c];
                const unsigned long long m0 = __ballot(static_cast<int>(excl) <= rr && rr < static_cast<int>(incl[c]));
                const unsigned long long m1 = __ballot(static_cast<int>(excl) <= rr + 1 && rr + 1 < static_cast<int>(incl[c]));
                if (m0 == 0ull || (want_next && m1 == 0ull)) {    // inconsistent counts (NaN in the range?): general path
                    live[c] = false;
                } else {
                    tb[c] = static_cast<uint32_t>(__builtin_ctzll(m0));
                    tb1[c] = want_next ? static_cast<uint32_t>(__builtin_ctzll(m1)) : tb[c];
                    const int first = __builtin_amdgcn_readlane(static_cast<int>(incl[c] - cnt4[c]), static_cast<int>(tb[c]));
                    const int last = __builtin_amdgcn_readlane(static_cast<int>(incl[c]), static_cast<int>(tb1[c]));
                    n_in[c] = last - first;                       // values in buckets tb .. tb1
                    if (n_in[c] <= 64) {
                        extract[c] = true;
                        below[c] += first;
                    } else if (level == 0 && tb1[c] == tb[c]) {   // split this bucket once more
                        below[c] += first;
                        plo[c] = lo[c];
                        pinv[c] = inv[c];
                        pb[c] = tb[c];
                        const float w = 1.0f / inv[c];             // ~ bucket width
                        lo[c] = uniform(plo[c] + (static_cast<float>(tb[c]) - 0.01f) * w);
                        inv[c] = uniform(static_cast<float>(kBuckets) * (1.0f - 1.0f / 1048576.0f) / (1.02f * w));
                        if (!finite_f(inv[c]) || !finite_f(lo[c])) live[c] = false;
                    } else {
                        live[c] = false;                          // too crowded: general path
                    }
                }
            }
        }
        // ---- compaction of the target buckets and one 64-lane sort for all columns
        bool any_extract = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) any_extract = any_extract || extract[c];
        if (any_extract) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (extract[c]) {
                    int n_cand = 0;
#pragma unroll
                    for (int g = 0; g < RPL / GS; ++g) {
                        if (g < groups) {
#pragma unroll
                            for (int jj = 0; jj < GS; ++jj) {
                                const float a = mag<ABS>(x[c][g * GS + jj]);
                                uint32_t b = bucket_of(a, lo[c], inv[c]);
                                bool in;
                                if (level == 0) {
                                    in = b >= tb[c] && b <= tb1[c];
                                } else {
                                    b = b < kBuckets ? b : kBuckets - 1;
                                    in = bucket_of(a, plo[c], pinv[c]) == pb[c] && b >= tb[c] && b <= tb1[c];
                                }
                                const unsigned long long m = __ballot(in);
                                if (m) {
                                    const int pos = n_cand + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                                    if (in && pos < 64) cand[c * 64 + pos] = a;
                                    n_cand += __popcll(m);
                                }
                            }
                        }
                    }
                    if (n_cand != n_in[c]) extract[c] = false;   // the histogram and the sweep disagree: general path
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float sv[NC][1];
#pragma unroll
            for (int c = 0; c < NC; ++c) sv[c][0] = (extract[c] && lane < n_in[c]) ? cand[c * 64 + lane] : pinf;
            wave_bitonic_sort<1, NC>(sv, lane);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (extract[c]) {
                    const int i = r - below[c];
                    res_a[c] = lane_value(sv[c][0], i & 63);
                    res_b[c] = want_next ? lane_value(sv[c][0], (i + 1) & 63) : res_a[c];
                    ok[c] = true;
                    live[c] = false;
                }
            }
        }
    }
}

// Ties at exactly t beyond the keep-th value (rare with continuous data, normal when many clients submit the
// same vector): the reference's stable sort keeps the lowest rows.  v holds the deviations.
template <int RPL>
__device__ __forceinline__ float tied_window_sum(const float (&v)[RPL], int groups, int keep, float t) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;
    float acc = 0.0f;
    int n_closer = 0, n_pos = 0, n_neg = 0;
#pragma unroll
    for (int g = 0; g < RPL / GS; ++g) {
        if (g < groups) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) {
                const float d = v[g * GS + jj];
                const bool closer = __builtin_fabsf(d) < t;
                acc += closer ? d : 0.0f;
                n_closer += __popcll(__ballot(closer));
                n_pos += __popcll(__ballot(d == t));
                n_neg += __popcll(__ballot(d == -t));
            }
        }
    }
    float sum = wave_sum(acc);
    const int need = keep - n_closer;
    if (t == 0.0f) {
        // ties are zeros of either sign: they add nothing
    } else if (need == n_pos + n_neg) {
        sum += static_cast<float>(n_pos - n_neg) * t;
    } else if (n_neg == 0) {
        sum += static_cast<float>(need) * t;
    } else if (n_pos == 0) {
        sum -= static_cast<float>(need) * t;
    } else {
        int taken = 0, pos_taken = 0, neg_taken = 0;
#pragma unroll
        for (int g = 0; g < RPL / GS; ++g) {
            if (g < groups) {
#pragma unroll
                for (int jj = 0; jj < GS; ++jj) {  // register j holds rows 64 j + lane: ascending row order
                    const float d = v[g * GS + jj];
                    const bool tied = __builtin_fabsf(d) == t;
                    const unsigned long long m = __ballot(tied);
                    const int before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                    const bool take = tied && (taken + before < need);
                    pos_taken += __popcll(__ballot(take && d > 0.0f));
                    neg_taken += __popcll(__ballot(take && d < 0.0f));
                    taken += __popcll(m);
                }
            }
        }
        sum += static_cast<float>(pos_taken - neg_taken) * t;
    }
    return sum;
}

// Everything after the tile's values sit in registers: NaN screening, the two selections, the window sum, the store.
template <int RPL, int NC>
__device__ __forceinline__ void finish_tile(float (&x)[NC][RPL], const float (&mn)[NC], const float (&mx)[NC],
                                            bool suspicious, int chunks, int slots, int n_rows, int keep, int lane,
                                            int wave, float* strip, int64_t c_base, int64_t n_cols,
                                            float* __restrict__ out) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;
    const float pinf = __builtin_inff();
    const float qnan = __uint_as_float(0x7fc00000u);
    Bracket q[4];
    bool dead[NC];       // the column's result is NaN
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        dead[c] = keep <= 0;   // np.mean([]) is nan
        if (suspicious) {
            int n_nan = 0;
#pragma unroll
            for (int g = 0; g < RPL / GS; ++g) {
                if (g < chunks) {
#pragma unroll
                    for (int jj = 0; jj < GS; ++jj) {
                        const float e = x[c][g * GS + jj];
                        n_nan += __popcll(__ballot(e != e));
                    }
                }
            }
            dead[c] = dead[c] || n_nan > 0;   // a NaN in the column makes np.median nan
        }
    }

    // ---- median (defences.py:49)
    float med[NC], lo_x[NC], hi_x[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        lo_x[c] = wave_min(mn[c]);
        hi_x[c] = wave_max(mx[c]);
        q[c].hi = hi_x[c];
        q[c].c_hi = n_rows;
        q[c].c_lo = 0;
        q[c].lo = uniform(from_fkey(fkey(lo_x[c]) - 1u));   // the float just below the minimum
        q[c].a = q[c].b = 0.0f;
        if (!(lo_x[c] > -pinf) || dead[c]) {   // -inf in the data (or nothing to do): bracket by key space alone
            q[c].lo = -pinf;
            q[c].c_lo = 0;
            if (!dead[c]) {
                int c_inf = 0;
#pragma unroll
                for (int g = 0; g < RPL / GS; ++g)
                    if (g < chunks)
#pragma unroll
                        for (int jj = 0; jj < GS; ++jj) c_inf += __popcll(__ballot(x[c][g * GS + jj] == -pinf));
                q[c].c_lo = c_inf;
            }
        }
    }
    const bool even = (n_rows & 1) == 0;
    const int r_med = even ? (n_rows >> 1) - 1 : (n_rows - 1) >> 1;
    bool run_med[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        // more -inf than the median rank: the median is -inf and every deviation is NaN or +inf
        run_med[c] = !dead[c] && q[c].c_lo <= r_med;
        if (!run_med[c]) {
            dead[c] = true;
            q[c].lo = 0.0f;   // harmless bracket: the column is ignored
            q[c].hi = 0.0f;
            q[c].c_lo = 0;
            q[c].c_hi = 0;
        }
    }
#pragma unroll
    for (int c = NC; c < 4; ++c) q[c] = Bracket{0.0f, 0.0f, 0, 0, 0.0f, 0.0f};
    {
        bool want[NC], got[NC];
        float ba[NC], bb[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) want[c] = run_med[c] && lo_x[c] > -pinf;
        // (the one-column-per-wave variant has no registers to spare for the bucket bookkeeping: probing search only)
        if constexpr (NC == 4 && RPL <= kBucketMaxRpl) bucket_select<RPL, NC, false>(x, chunks, r_med, even, lo_x, hi_x, want, ba, bb, got, lane, strip);
        else for (int c = 0; c < NC; ++c) got[c] = false;
        bool generic = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) generic = generic || (run_med[c] && !got[c]);
        if (generic) {
            select4<RPL, NC, false>(x, chunks, slots, r_med, even, q, lane, strip);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                q[c].a = run_med[c] ? ba[c] : 0.0f;
                q[c].b = run_med[c] ? bb[c] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) med[c] = uniform(even ? __fmul_rn(__fadd_rn(q[c].a, q[c].b), 0.5f) : q[c].a);

    // ---- deviations in place (defences.py:50: column - med); rounding is monotone, so the largest
    // |deviation| belongs to one of the extremes.  Padding stays +inf.
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int g = 0; g < RPL / GS; ++g) {
            if (g < chunks) {
#pragma unroll
                for (int jj = 0; jj < GS; ++jj) x[c][g * GS + jj] = __fsub_rn(x[c][g * GS + jj], med[c]);
            }
        }
        const float max_dev = uniform(__builtin_fmaxf(__builtin_fabsf(__fsub_rn(lo_x[c], med[c])),
                                                      __builtin_fabsf(__fsub_rn(hi_x[c], med[c]))));
        dead[c] = dead[c] || !(max_dev == max_dev);   // inf - inf
        q[c].lo = -__uint_as_float(1u);               // below every |d|
        q[c].c_lo = 0;
        q[c].hi = max_dev;
        q[c].c_hi = n_rows;
        if (dead[c]) {
            q[c].lo = 0.0f;
            q[c].hi = 0.0f;
            q[c].c_hi = 0;
        }
    }
    // ---- t = the keep-th smallest |deviation|
    {
        bool want[NC], got[NC];
        float ba[NC], bb[NC], zero[NC], top[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            want[c] = !dead[c];
            zero[c] = 0.0f;
            top[c] = q[c].hi;   // max |deviation|
        }
        if constexpr (NC == 4 && RPL <= kBucketMaxRpl) bucket_select<RPL, NC, true>(x, chunks, keep - 1, false, zero, top, want, ba, bb, got, lane, strip);
        else for (int c = 0; c < NC; ++c) got[c] = false;
        bool generic = false;
#pragma unroll
        for (int c = 0; c < NC; ++c) generic = generic || (!dead[c] && !got[c]);
        if (generic) {
            select4<RPL, NC, true>(x, chunks, slots, keep - 1, false, q, lane, strip);
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) q[c].a = dead[c] ? 0.0f : ba[c];
        }
    }

    // ---- window sum: everything with |d| <= t, when that is exactly `keep` values
    float result[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float t = q[c].a;
        float acc = 0.0f;
        int n_in = 0;
        if (!dead[c]) {
#pragma unroll
            for (int g = 0; g < RPL / GS; ++g) {
                if (g < chunks) {
#pragma unroll
                    for (int jj = 0; jj < GS; ++jj) {
                        const float d = x[c][g * GS + jj];
                        const bool in = __builtin_fabsf(d) <= t;
                        acc += in ? d : 0.0f;
                        n_in += __popcll(__ballot(in));
                    }
                }
            }
        }
        float sum = wave_sum(acc);
        if (!dead[c] && n_in != keep) sum = tied_window_sum<RPL>(x[c], chunks, keep, t);
        // defences.py:51: np.mean(good) + med
        result[c] = dead[c] ? qnan : __fadd_rn(__fdiv_rn(sum, static_cast<float>(keep)), med[c]);
    }
    if (lane < NC) {
        const int64_t col = c_base + NC * wave + lane;
        float r = result[0];
        if constexpr (NC == 4) r = lane == 0 ? result[0] : (lane == 1 ? result[1] : (lane == 2 ? result[2] : result[3]));
        if (col < n_cols) out[col] = r;
    }
}

// NC columns per wave, WAVES waves per workgroup, WAVES * NC = 16 columns per tile.  NC = 4 (4 waves) holds up to
// 2560 rows; NC = 1 (16 waves, one column each) trades the amortisation of the probe bookkeeping for register space
// and holds up to 5632 rows (Bulyan's second stage at N = 10,000: theta = 5200).
// MODE 0: the general selection for every tile.  MODE 2: the general selection for the tiles listed in `redo` -- launched
// right behind the ring selection of window_lean.hip, which resolves all but a few tiles in a thousand and lists the rest.
// (MODE 1 was this file's own ring selection over the column-split layout, round 2's fast path up to 1024 rows, and
// window_rows.hip its row-split form above that: both were superseded by window_lean.hip in round 3 -- 0.64 / 0.56 -> 0.40 ms
// at 1000 rows, 1.21 -> 0.94 ms at 2080 rows per 2^18 columns -- and removed.)

// (A branch-free form of the staging loads -- every load of a chunk unconditional, padding rows re-reading the last row --
// was built at the end of round 2 on the strength of the ISA (s_waitcnt vmcnt(0) in front of every guarded load) and measured
// in round 3: bitwise the same results, 1.256 ms against 0.648 ms at 1000 rows x 2^18 columns
// (profiles/r03a_optin_variants_probe.txt).  It was removed; the guarded loads stay.)
template <int RPL, int NC, int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES, (RPL * NC <= 64 || WAVES == 16 ? 4 : 2)) void median_window_kernel(
    const float* __restrict__ G, int n_rows, int64_t n_cols, int64_t ld, const int32_t* __restrict__ row_index,
    int keep, float* __restrict__ out, int32_t* __restrict__ redo) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;                    // registers (64-row groups) per guard group
    constexpr int JC = GS;   // registers per transit chunk (512-row chunks cost a wave per SIMD of occupancy: 2.38 -> 2.96 ms)
    constexpr int NCH = RPL / JC;
    constexpr int COLS = WAVES * NC;          // columns per tile
    constexpr int STRIDE = COLS + 4;          // floats per LDS row: 16-byte aligned, conflict-free b128 column reads
    constexpr int QUADS = COLS / 4;
    constexpr int THREADS_ = 64 * WAVES;
    constexpr int ROWS_PER_PASS = THREADS_ / QUADS;   // rows one staging pass of the workgroup covers
    constexpr int kTransit = 64 * JC * STRIDE;
    // per-wave scratch behind the transit buffer: the general path's strip/histogram/candidates, or the ring path's
    // histogram (its gather lists reuse the transit buffer, dead once the tile sits in registers)
    constexpr int kScratch = kWaveScratch;
    __shared__ __attribute__((aligned(16))) float transit[kTransit + WAVES * kScratch + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* strip = transit + kTransit + wave * kScratch;
    int* nonfinite = reinterpret_cast<int*>(transit + kTransit + WAVES * kScratch);   // one flag per column quad
    // (an XCD-contiguous tile order, so that the two tiles sharing a 128-byte line run on one XCD, measured no effect)
    int64_t tile = blockIdx.x;
    if constexpr (MODE == 2) {
        if (static_cast<int>(blockIdx.x) >= redo[0]) return;   // redo[0] = number of listed tiles, redo[1 + i] = tile i
        tile = redo[1 + blockIdx.x];
    }
    const int64_t c_base = tile * COLS;
    const float pinf = __builtin_inff();
    const float qnan = __uint_as_float(0x7fc00000u);
    const int chunks = (n_rows + 64 * JC - 1) / (64 * JC);   // transit chunks to stage
    const int groups = (n_rows + 64 * GS - 1) / (64 * GS);   // guard groups in use
    const int slots = groups * 64 * GS;

    if (tid < QUADS) nonfinite[tid] = 0;

    // ---- stage: global (128-byte row segments) -> LDS transit -> registers (4 columns x RPL rows per lane)
    const int ld_q = (tid % QUADS) * 4, ld_r = tid / QUADS;
    const int64_t ld_c = c_base + ld_q;
    constexpr int PASSES = 64 * JC / ROWS_PER_PASS;   // float4 loads per thread per chunk
    f32x4 tmp[PASSES];
    float poison = 0.0f;   // x * 0 accumulates to NaN as soon as one loaded value is NaN or +-inf
    auto fetch = [&](int ch) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = ch * 64 * JC + ROWS_PER_PASS * p + ld_r;
            f32x4 val = {pinf, pinf, pinf, pinf};
            if (row < n_rows) {
                const int64_t src = row_index ? row_index[row] : row;
                const float* ptr = G + src * ld + ld_c;
                if (ld_c + 4 <= n_cols) {
                    val = *reinterpret_cast<const f32x4u*>(ptr);
                } else {  // ragged last tile: columns past the matrix are computed on zeros and never stored
                    val = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    if (ld_c + 0 < n_cols) val.x = ptr[0];
                    if (ld_c + 1 < n_cols) val.y = ptr[1];
                    if (ld_c + 2 < n_cols) val.z = ptr[2];
                }
            }
            tmp[p] = val;
        }
    };
    auto stash = [&](int ch) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            f32x4 val = tmp[p];
            const bool real_row = ch * 64 * JC + ROWS_PER_PASS * p + ld_r < n_rows;
            if (real_row)
                poison = __builtin_fmaf(val.x + val.y, 0.0f, __builtin_fmaf(val.z + val.w, 0.0f, poison));
            *reinterpret_cast<f32x4*>(transit + (ROWS_PER_PASS * p + ld_r) * STRIDE + ld_q) = val;
        }
    };
    float x[NC][RPL];
    float mn[NC], mx[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        mn[c] = pinf;
        mx[c] = -pinf;
    }
    fetch(0);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch < chunks) {
            stash(ch);
            if (ch + 1 < chunks) fetch(ch + 1);
            __syncthreads();
            const bool last_chunk = ch == chunks - 1;   // uniform: only this chunk can hold padding rows
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                float val[NC];
                if constexpr (NC == 4) {
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(transit + (64 * jj + lane) * STRIDE + 4 * wave);
                    val[0] = v4.x; val[1 % NC] = v4.y; val[2 % NC] = v4.z; val[3 % NC] = v4.w;
                } else {
                    val[0] = transit[(64 * jj + lane) * STRIDE + wave];
                }
                const bool real = !last_chunk || (ch * 64 * JC + 64 * jj + lane < n_rows);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float e = val[c];
                    x[c][ch * JC + jj] = e;
                    mn[c] = __builtin_fminf(mn[c], e);                 // +inf padding never wins a minimum
                    mx[c] = __builtin_fmaxf(mx[c], real ? e : -pinf);  // and is masked out of the maximum
                }
            }
            __syncthreads();
        }
    }
    if (poison != poison) nonfinite[tid % QUADS] = 1;
    __syncthreads();
    const bool suspicious = uniform(nonfinite[NC == 4 ? wave : wave / 4]) != 0;   // an LDS load is per-lane to the compiler: make it scalar

    finish_tile<RPL, NC>(x, mn, mx, suspicious, groups, slots, n_rows, keep, lane, wave, strip, c_base, n_cols, out);
}

template <int RPL, int NC, int WAVES>
int launch_rpl(const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index, int64_t keep,
               float* out, hipStream_t stream) {
    const int64_t n_tiles = ceil_div(n_cols, static_cast<int64_t>(WAVES * NC));
    median_window_kernel<RPL, NC, WAVES, 0><<<static_cast<unsigned>(n_tiles), 64 * WAVES, 0, stream>>>(
        G, static_cast<int>(n_rows), n_cols, ld, row_index, static_cast<int>(keep), out, nullptr);
    return check_launch("median_window_kernel");
}

// The ring selection of window_lean.hip first, then this file's general kernel over whatever it could not resolve (redo[0]
// tiles; the second launch is sized for all of them and the surplus workgroups leave at once).
template <int RPL, int NC, int WAVES>
int launch_ring(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                int64_t keep, float* out, hipStream_t stream) {
    const int64_t n_tiles = ceil_div(n_cols, static_cast<int64_t>(kCols));
    BYZ_TRY(ctx->redo_tiles.ensure(static_cast<size_t>(n_tiles + 1) * sizeof(int32_t)));
    int32_t* redo = ctx->redo_tiles.as<int32_t>();
    BYZ_HIP(hipMemsetAsync(redo, 0, sizeof(int32_t), stream));
    const int rc = launch_window_lean(ctx, G, n_rows, n_cols, ld, row_index, keep, out, redo, stream);
    if (rc == BYZ_E_UNSUPPORTED) {   // (a leading dimension of 2^30 elements or more)
        ctx->redo_valid = false;
        return launch_rpl<RPL, NC, WAVES>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    }
    if (rc != BYZ_OK) return rc;
    median_window_kernel<RPL, NC, WAVES, 2><<<static_cast<unsigned>(n_tiles), 64 * WAVES, 0, stream>>>(
        G, static_cast<int>(n_rows), n_cols, ld, row_index, static_cast<int>(keep), out, redo);
    return check_launch("median_window_kernel<redo>");
}

}  // namespace

int launch_trimmed_mean_sorted(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                               const int32_t* row_index, int64_t keep, float* out, hipStream_t stream);

int launch_trimmed_mean(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                        const int32_t* row_index, int64_t keep, float* out, hipStream_t stream) {
    BYZ_REQUIRE(G && out && n_rows > 0 && n_cols > 0 && ld >= n_cols, "trimmed_mean: bad shape %lld x %lld ld %lld",
                (long long)n_rows, (long long)n_cols, (long long)ld);
    BYZ_REQUIRE(keep >= 0 && keep <= n_rows, "trimmed_mean: keep count %lld out of range", (long long)keep);
    BYZ_REQUIRE(ceil_div(n_cols, kCols) <= 0x7fffffff, "trimmed_mean: too many columns");
    KernelTimer t(ctx, BYZ_K_TRIMMED_MEAN, stream);
    ctx->redo_valid = false;
    {
        // beyond the register kernels' 5,632 rows: a radix select over the column streamed from HBM (tall_select.hip); BYZ_TM_LARGE=1
        // forces the global-memory sort at every height, BYZ_TM_TALL=0 leaves the heights to the two sorts of rounds 3-6
        const char* forced_sort = std::getenv("BYZ_TM_LARGE");
        const bool sort_anyway = forced_sort != nullptr && std::atoi(forced_sort) != 0;
        if (!sort_anyway && trimmed_mean_tall_applies(n_rows))
            return launch_trimmed_mean_tall(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
    }
    if (trimmed_mean_large_applies(n_rows)) return launch_trimmed_mean_large(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
    const int64_t rpl = ceil_div(n_rows, 64);
    if (rpl > 88) return launch_trimmed_mean_sorted(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
    // the ring selection (window_lean.hip) with this file's general kernel behind it: 129 .. 5376 rows
    // (BYZ_TM_RING=0 keeps the general kernel for everything: the comparison, and tests of the general kernel itself)
    {
        const char* e = std::getenv("BYZ_TM_RING");
        if ((!e || std::atoi(e) != 0) && keep >= 1 && rpl >= 3 && rpl <= 84) {
            ctx->redo_valid = true;
            if (rpl <= 4) return launch_ring<4, 4, 4>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
            if (rpl <= 8) return launch_ring<8, 4, 4>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
            if (rpl <= 16) return launch_ring<16, 4, 4>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
            if (rpl <= 24) return launch_ring<24, 4, 4>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
            if (rpl <= 32) return launch_ring<32, 4, 4>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
            if (rpl <= kMaxRpl) return launch_ring<40, 4, 4>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
            if (rpl <= 64) return launch_ring<64, 1, 16>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
            return launch_ring<88, 1, 16>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
        }
    }
    if (rpl <= 1) return launch_rpl<1, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 2) return launch_rpl<2, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 4) return launch_rpl<4, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 8) return launch_rpl<8, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 16) return launch_rpl<16, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 24) return launch_rpl<24, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 32) return launch_rpl<32, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= kMaxRpl) return launch_rpl<40, 4, 4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 64) return launch_rpl<64, 1, 16>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    return launch_rpl<88, 1, 16>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
}

}  // namespace byz
