// Median-window trimmed mean (reference defences.py:44-52), row-split layout, round 3: the instruction-lean form.
//
// window_rows.hip (round 2; removed when this file replaced it) proved the layout -- the workgroup owns 16 columns, the waves split the rows, every load of
// the tile in flight at once -- and its counters showed what bounds it: 65 vector instructions per value (1.05e9 VALU
// wave-instructions per 1e9 values, profiles/r02k), i.e. instruction issue, not HBM.  Half of them were the four sweeps
// over the register-resident tile, half the per-column bookkeeping done by every wave with all 64 lanes for one column at
// a time.  This kernel keeps the selection rule (rings of equal-width buckets around the median buckets; see below) and
// removes instructions:
//
//   sweep A   histogram only.  No min / max sweep: the bucket range comes from a SAMPLE of the rows (three register rows per
//             lane, spread over the tile: 48 W rows) widened by an eighth on both sides; values outside land in the two end
//             buckets, and a column whose window could reach an end bucket is not resolved here (launcher: columns that
//             keep more than 90% of their rows take the EXACT instantiation, which sweeps for the true range).  The bucket
//             is round(x * inv + nlo), formed by ONE v_pk_fma_f32 per two values with 2^23 folded into the addend (the
//             sum's mantissa IS the bucket), clamped by one v_med3_f32; counters are 16 bits wide and two columns share a
//             word, the half chosen by the column's parity -- a compile-time constant, so the address is one v_lshl_add.
//             2.5 vector instructions + one LDS atomic per value (round 2: 3 + 7..10).
//   scan      the waves turn the histograms into exclusive prefix sums, two columns per add (packed halves), DPP scans.
//   search    ONE wave, one lane per column: bucket of the median ranks, the ring j* that completes `keep` values and the
//             counts that go with it, by branch-free binary searches on the prefix sums -- ~500 instructions per TILE
//             (round 2: ~1300 per wave, with all lanes on one column at a time).
//   sweep B   gather + sum in one pass: ring(x) = max(bm1 - b, b - bm2) from two packed subtractions and a max;
//             0 < ring < ring_lo: decided in, summed as (x - pivot) right here; median buckets and undecided rings: pushed
//             on the lane's stack.  ~11 instructions per value (round 2: gather 13 + a separate sum sweep 6).
//   owners    per column: compact the stacks, one 64-lane sort by value (the median exactly as np.median forms it), one
//             bitonic merge of |fl(x - med)| (the threshold), the kept candidates summed as (x - pivot) in sorted order.
//
// The selection rule (scripts/proto/ring_window.py is its numpy model).  u(x) = x * inv + nlo is affine and increasing,
// bucket b holds u in [b - 1/2, b + 1/2].  With bm1 <= bm2 the buckets of the median ranks and s = 1 + bm2 - bm1, a value
// of ring j = max(bm1 - b, b - bm2) deviates from the median by delta in [j - 1, j + s] (in units of u).  j* = the first
// ring with N(j*) >= keep values inside; the keep-th smallest deviation D then lies in [j* - 1, j* + s].  So rings
// <= j* - 1 - s are inside the window (delta <= j* - 1 <= D), rings >= j* + s + 2 are outside with a whole bucket to
// spare (delta >= j* + s + 1 > D: no tie with the window's edge is possible, which is what lets this kernel do without
// round 2's recount of all values), and rings j* - s .. j* + s + 1 are undecided: they are gathered and decided exactly.
// Ties AT the edge among the gathered values (more than `need` of them within the threshold) are the general kernel's
// business, as before, and so is everything else this cannot resolve: the tile goes on the redo list.
//
// The result is sum(x - pivot) / keep + pivot with pivot = the centre of the median bucket, where the reference forms
// mean(x - med) + med: the same number up to the rounding of the individual differences (tolerance 1e-5, north_star).
// Bound: HBM, 4 R D + 4 D bytes.
#include "common.hpp"

#include "lane_exchange.hpp"

#include <cstdlib>

namespace byz {
namespace {

using namespace lanes;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kTileCols = 16;
constexpr float kMagic = 8388608.0f;   // 2^23: a float in [2^23, 2^24) has ulp 1 -- adding it rounds to an integer

__device__ __forceinline__ uint32_t okey(float v) {   // order-preserving float -> uint32
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_okey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ bool is_finite(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }
// Sweep B keeps its predicates as LANE MASKS (what v_cmp writes: an SGPR pair) instead of per-lane booleans (round 6).  From
// `acc += in ? d : 0; top += hit ? 1 : 0` hipcc makes two v_cndmask and a v_addc per value: it folds two steps' increments
// into one add-with-carry and materialises the other step's as a 0 / 1 register.  With the masks in hand the increment is ONE
// v_addc (carry-in = the mask) and `hit = le & ~in` is scalar work.
typedef unsigned long long lane_mask;
__device__ __forceinline__ lane_mask lanes_lt(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 4); }   // ordered <
__device__ __forceinline__ lane_mask lanes_le(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 5); }   // ordered <=
__device__ __forceinline__ int add_mask(int count, lane_mask m) {      // count + (lane in m ? 1 : 0)
    lane_mask carry_out;
    int out;
    asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(out), "=s"(carry_out) : "v"(count), "s"(m));
    return out;
}
__device__ __forceinline__ float keep_mask(float v, lane_mask m) {     // lane in m ? v : +0.0
    float out;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(out) : "v"(v), "s"(m));
    return out;
}
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// lanes q, q + 4, q + 8, q + 12 of a 16-lane row hold the same column quad: two rotations inside the row combine them
__device__ __forceinline__ float row_quad_min(float v) {
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true)));   // row_ror:4
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true)));   // row_ror:8
    return v;
}
__device__ __forceinline__ float row_quad_max(float v) {
    v = __builtin_fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true)));
    v = __builtin_fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true)));
    return v;
}
__device__ __forceinline__ float row_quad_sum(float v) {   // fixed order: (l + l-4) + (l-8 + l-12)
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true)));
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true)));
    return v;
}

// inclusive scan of 32-bit words over groups of G = 32 or 64 consecutive lanes (packed 16-bit halves add independently)
template <int G>
__device__ __forceinline__ uint32_t group_scan(uint32_t v) {
    int x = static_cast<int>(v);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1 (zeros shifted in)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8: every 16-lane row is scanned
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    if constexpr (G == 64) x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2, 3
    return static_cast<uint32_t>(x);
}

// sum over the 64 lanes in a fixed order, no LDS trip: four DPP steps inside each 16-lane row, then the row totals in order
__device__ __forceinline__ float wave_sum_f(float v, int lane) {
    (void)lane;
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));   // row_half_mirror
    v = __fadd_rn(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));   // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return __fadd_rn(__fadd_rn(r0, r1), __fadd_rn(r2, r3));
}
__device__ __forceinline__ int wave_sum_i(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);   // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
           __builtin_amdgcn_readlane(x, 48);
}

// why tiles went to the general kernel (development aid: BYZ_TM_LEAN_DEBUG=1 makes the launcher print and reset them):
//  0 non-finite input / keep < 1   1 degenerate or too fine a range   2 median ranks buckets apart   3 more candidates than
//  the sort takes   4 need out of range   5 a ring touches an end bucket of the sampled range   6 a lane's stack overflowed
//  7 gather and histogram disagree   8 ties at the window's edge   9 threshold NaN
__device__ unsigned int g_lean_reasons[16];
// development aid (BYZ_TM_LEAN_TIMING=1): s_memtime stamps of the phases of the first 256 tiles, taken by thread 0
constexpr int kStampTiles = 256, kStamps = 12;
__device__ unsigned long long g_lean_stamps[kStampTiles * kStamps];

// Register rows are loaded (and then swept) in this order: the three sampled ones first, so that the range phase can start
// while the rest of the tile is still on its way (memory returns loads in order).
template <int RPW>
__host__ __device__ constexpr int lean_order(int i) {
    const int s1 = RPW / 3, s2 = (2 * RPW) / 3;
    if (i == 0) return 0;
    if (i == 1) return s1;
    if (i == 2) return s2;
    int k = i - 3;
    for (int j = 0; j < RPW; ++j) {
        if (j == 0 || j == s1 || j == s2) continue;
        if (k == 0) return j;
        --k;
    }
    return 0;
}

struct ColumnPlan {   // written by the search (the first lane of the column's group), read by sweep B and by the column's owner
    // sweep B classifies a value by z = 2 b - (bm1 + bm2), b its bucket: |z| = s - 1 + 2 ring, s = 1 + bm2 - bm1
    float sum_b;         // bm1 + bm2
    float mid, half;     // decided in (rings 1 .. ring_lo - 1)  <=>  | |z| - mid | < half;  half = -1: none
    float outer;         // gathered unless decided in  <=>  |z| <= outer;  -1: the column gathers nothing (unresolved)
    float pivot;         // centre of the first median bucket: the sums run over x - pivot
    int expected;        // values the gather must find
    int need;            // undecided values that belong to the window
    int idx1, idx2;      // positions of the median ranks among the sorted gathered values
    int marked, mid_lo;  // median-bucket values gathered for the median only (decided in): count, first position
};

// W waves, RPW 16-row blocks per wave (rows <= 16 W RPW), B buckets, SR sort registers per lane (64 SR candidates per column),
// LS stack slots per lane and column.  EXACT: the bucket range is the column's true [min, max] (one more sweep); otherwise
// it is taken from a sample and the end buckets collect what falls outside.
template <int W, int RPW, int B, int SR, int LS, bool EXACT, bool PIPE>
__global__ __launch_bounds__(64 * W, 4) void window_lean_kernel(const float* __restrict__ G, int n_rows, int64_t n_cols,
                                                                int64_t ld, const int32_t* __restrict__ row_index, int keep,
                                                                float* __restrict__ out, int32_t* __restrict__ redo,
                                                                int by_xcd, int stamp_from) {
    constexpr int T = 64 * W;
    constexpr int NCW = kTileCols / W;          // columns an owner wave resolves: 4, 2 or 1
    constexpr int CAP = 64 * SR;
    constexpr int SCAN_WAVES = W < 8 ? W : 8;   // waves that scan histograms: 8 column pairs in all
    constexpr int CPS = 8 / SCAN_WAVES;         // column pairs per scanning wave: 2 (W = 4) or 1
    constexpr int GRP = 64 / CPS;               // lanes per column pair
    static_assert(B % (4 * GRP) == 0, "scan layout");
    constexpr int PS = B + 4;                   // words per column pair: B buckets + entry B (the column's total) + padding
    static_assert(B == 512 || B == 1024 || B == 2048, "the search below is written for these");
    // un[]: 8 x PS words of histogram / prefix sums (two columns per word), then 4 (LS + 1) T words of gather stacks
    extern __shared__ __attribute__((aligned(16))) uint32_t un[];
    __shared__ float dense[kTileCols * CAP];
    __shared__ uint32_t tops[T];
    __shared__ uint32_t minmax[2 * kTileCols];
    __shared__ ColumnPlan plan[kTileCols];
    __shared__ float part_sum[W * kTileCols];
    __shared__ float colres[kTileCols];
    __shared__ int flags[2];   // [0] non-finite input seen, [1] the tile is not resolved here

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int rr = lane >> 2, q = lane & 3;
    // Workgroup -> tile.  A tile's rows are 64-byte segments, half a 128-byte line each, and consecutive workgroups go to
    // different XCDs (each with its own L2): with tile = blockIdx.x the two halves of every line are fetched by two L2s at
    // different times.  Instead XCD x (blockIdx.x & 7, an affinity the dispatcher follows but does not promise: only speed
    // depends on it) takes the x-th eighth of the tiles in order, so that neighbouring tiles are neighbours in time on ONE L2.
    // Same-box A/B (scripts/tm_ab.py BYZ_TM_LEAN_XCD 0,1): 1000 rows 0.375 -> 0.354 ms per 2^18 columns, 2080 rows 0.471 ->
    // 0.410 per 2^17, 5200 rows 0.766 -> 0.625 per 2^16; groups of 2 / 4 / 16 / 64 tiles per XCD instead of eighths: worse or equal.
    int64_t tile = blockIdx.x;
    if (by_xcd != 0) {
        const int64_t per = gridDim.x >> 3;
        if (tile < (per << 3)) tile = (tile & 7) * per + (tile >> 3);
    }
    const int64_t c_base = tile * kTileCols;
    const float pinf = __builtin_inff();
    // stamp_from = 1 + first stamped tile (0: off).  A kernel argument: as a flag in device memory it was a dependent scalar load
    // that every workgroup waited for before its first instruction of substance.
    const int64_t stamp_tile = tile - (stamp_from - 1);
    const bool stamping = stamp_from != 0 && stamp_tile >= 0 && stamp_tile < kStampTiles && tid == 0;
#define BYZ_STAMP(i) do { if (stamping) g_lean_stamps[stamp_tile * kStamps + (i)] = __builtin_readcyclecounter(); } while (0)
    BYZ_STAMP(0);

    // ---- loads.  No branch around a load (rows past the matrix re-read the last row and are masked where they are used).
    // A wave cannot get past its load instructions while the CU's memory pipeline is full -- measured: the 16 loads of a
    // 1000-row tile hold wave 0 for 7,000 of the workgroup's 37,000 ticks.  With several workgroups on the CU (4 and 8 waves
    // per tile) that is covered by the others, and requesting the whole tile at once keeps the most requests in flight:
    // requesting rows a few ahead of their use inside sweep A instead was measured SLOWER there (0.442 vs 0.400 ms at 1000
    // rows, 0.918 vs 0.881 ms at 2080).  The 16-wave shapes have the CU to themselves: there only the three SAMPLED register
    // rows are requested up front and the others under the range phase and the histogram (1.526 -> 1.409 ms at 5200 rows).
    // (EXACT needs every row for the range: everything up front.)
    // A ragged last tile (n_cols not a multiple of 16) is the general kernel's: its scalar, guarded loads have no place here.
    if (c_base + kTileCols > n_cols) {   // uniform
        if (tid == 0) redo[1 + atomicAdd(redo, 1)] = static_cast<int32_t>(tile);
        return;
    }
    f32x4 x[RPW];
    uint32_t src[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        int row = (j * W + wave) * 16 + rr;
        row = row < n_rows ? row : n_rows - 1;
        src[j] = static_cast<uint32_t>(row_index ? row_index[row] : row);
    }
    const unsigned char* const col_base = reinterpret_cast<const unsigned char*>(G) + (c_base + 4 * q) * 4;
    const uint32_t pitch = static_cast<uint32_t>(ld) * 4u;
    auto fetch = [&](int j) __attribute__((always_inline)) {
        x[j] = *reinterpret_cast<const f32x4u*>(col_base + static_cast<uint64_t>(src[j]) * pitch);
    };
    constexpr int PRE = (EXACT || !PIPE) ? RPW : 3;   // positions (in lean_order) requested before the range phase
    constexpr int AHEAD = 4;               // ... and how far sweep A requests ahead of what it counts
#pragma unroll
    for (int i = 0; i < PRE; ++i) fetch(lean_order<RPW>(i));
    BYZ_STAMP(8);   // (the first loads issued)
    // LDS set-up while the loads fly
    for (int i = tid; i < 8 * PS / 4; i += T) reinterpret_cast<uint4*>(un)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < kTileCols) {
        minmax[tid] = 0xffffffffu;               // min of ordered keys
        minmax[kTileCols + tid] = 0u;            // max
    }
    if (tid < 2) flags[tid] = 0;
    __syncthreads();
    BYZ_STAMP(1);   // loads issued, LDS cleared

    // ---- the bucket range: true minimum / maximum (EXACT) or those of a sample of the rows
    {
        float mn[4] = {pinf, pinf, pinf, pinf}, mx[4] = {-pinf, -pinf, -pinf, -pinf};
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            // the sample: three register rows spread over the tile (identical leading rows -- the attack's malicious clients --
            // must not be the whole sample)
            if (!EXACT && j != 0 && j != RPW / 3 && j != (2 * RPW) / 3) continue;
            const bool live = (j * W + wave) * 16 + rr < n_rows;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mn[e] = __builtin_fminf(mn[e], live ? x[j][e] : pinf);     // (rows past the matrix hold a copy of the last row)
                mx[e] = __builtin_fmaxf(mx[e], live ? x[j][e] : -pinf);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mn[e] = row_quad_min(mn[e]);
            mx[e] = row_quad_max(mx[e]);
        }
        if ((rr & 3) == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicMin(&minmax[4 * q + e], okey(mn[e]));
                atomicMax(&minmax[kTileCols + 4 * q + e], okey(mx[e]));
            }
        }
    }
    __syncthreads();
    BYZ_STAMP(2);   // (the first use of the loaded values: load latency ends here)

    // ---- sweep A: histograms.  bucket = round(x * inv + nlo) clamped to [0, B - 1]; column 4 q + e counts in the
    // low (e even) or high (e odd) half of word [column pair 2 q + e / 2][bucket]
    // (inv, nlo + 2^23) of a column: the same two numbers map a value to its bucket in both sweeps.  They are recomputed where
    // they are needed rather than kept: eight registers held across the phases were what pushed the tall shapes into scratch
    auto column_range = [&](int c, float& lo, float& hi) __attribute__((always_inline)) {
        lo = from_okey(minmax[c]);
        hi = from_okey(minmax[kTileCols + c]);
        if constexpr (!EXACT) {
            const float widen = 0.125f * (hi - lo);
            lo -= widen;
            hi += widen;
        }
    };
    auto column_scale = [&](float lo, float hi, float& iv, float& ka) __attribute__((always_inline)) {
        iv = static_cast<float>(B - 1) * __builtin_amdgcn_rcpf(hi - lo);   // (any monotone map will do: the same numbers everywhere)
        ka = __builtin_fmaf(-lo, iv, kMagic);
    };
    {
        float inv[4], kadd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float lo, hi;
            column_range(4 * q + e, lo, hi);
            column_scale(lo, hi, inv[e], kadd[e]);
        }
        f32x2 poison = {0.0f, 0.0f};   // x * 0 accumulates to NaN as soon as one live value is NaN or +-inf
        const f32x2 zero2 = {0.0f, 0.0f};
        const f32x2 inv01 = {inv[0], inv[1]}, inv23 = {inv[2], inv[3]};
        const f32x2 k01 = {kadd[0], kadd[1]}, k23 = {kadd[2], kadd[3]};
        // The clamped sum's BITS are 0x4B000000 + bucket, so (bits << 2) wraps to 0x2C000000 + 4 bucket: with that constant
        // folded into the histogram's LDS address the atomic's address is ONE v_lshl_add_u32 (no mask, no index arithmetic).
        typedef __attribute__((address_space(3))) uint32_t lds_word;
        const uint32_t un_at = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_word*)un));
        const uint32_t adj0 = un_at + static_cast<uint32_t>((2 * q) * PS * 4) - 0x2C000000u;        // column pair of e = 0, 1
        const uint32_t adj1 = un_at + static_cast<uint32_t>((2 * q + 1) * PS * 4) - 0x2C000000u;    // column pair of e = 2, 3
        uint32_t one_lo = 1u, one_hi = 65536u;
        asm volatile("" : "+v"(one_lo), "+v"(one_hi));   // two registers for the whole sweep (a constant operand is re-materialised per atomic)
        auto count = [&](const f32x4 v) __attribute__((always_inline)) {
            const f32x2 v01 = {v.x, v.y}, v23 = {v.z, v.w};
            const f32x2 y01 = __builtin_elementwise_fma(v01, inv01, k01);
            const f32x2 y23 = __builtin_elementwise_fma(v23, inv23, k23);
            poison = __builtin_elementwise_fma(v01, zero2, poison);
            poison = __builtin_elementwise_fma(v23, zero2, poison);
            const float hi_b = kMagic + static_cast<float>(B - 1);
            const uint32_t a0 = (__float_as_uint(__builtin_amdgcn_fmed3f(y01.x, kMagic, hi_b)) << 2) + adj0;
            const uint32_t a1 = (__float_as_uint(__builtin_amdgcn_fmed3f(y01.y, kMagic, hi_b)) << 2) + adj0;
            const uint32_t a2 = (__float_as_uint(__builtin_amdgcn_fmed3f(y23.x, kMagic, hi_b)) << 2) + adj1;
            const uint32_t a3 = (__float_as_uint(__builtin_amdgcn_fmed3f(y23.y, kMagic, hi_b)) << 2) + adj1;
            asm volatile("ds_add_u32 %0, %1" ::"v"(a0), "v"(one_lo) : "memory");
            asm volatile("ds_add_u32 %0, %1" ::"v"(a1), "v"(one_hi) : "memory");
            asm volatile("ds_add_u32 %0, %1" ::"v"(a2), "v"(one_lo) : "memory");
            asm volatile("ds_add_u32 %0, %1" ::"v"(a3), "v"(one_hi) : "memory");
        };
        auto request = [&](int i) __attribute__((always_inline)) {   // position i of the load order (nothing moves above here)
            if (i >= PRE && i < RPW) {
                const int j = lean_order<RPW>(i);
                asm volatile("" : "+v"(src[j]));
                fetch(j);
            }
        };
#pragma unroll
        for (int i = PRE; i < PRE + AHEAD; ++i) request(i);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            request(PRE + AHEAD + i);
            const int j = lean_order<RPW>(i);               // the order the loads were issued in
            if ((j * W + wave) * 16 + 15 < n_rows) {        // wave-uniform: a block without padding
                count(x[j]);
            } else if ((j * W + wave) * 16 + rr < n_rows) {
                count(x[j]);
            }
        }
        if (poison.x != poison.x || poison.y != poison.y) flags[0] = 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the atomics above are invisible to the compiler's counters
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {   // padding rows: +inf from here on (in no ring, in no sum)
        if ((j * W + wave) * 16 + 15 >= n_rows) {   // wave-uniform: only the last block or two can hold padding
            if ((j * W + wave) * 16 + rr >= n_rows) x[j] = f32x4{pinf, pinf, pinf, pinf};
        }
    }
    __syncthreads();
    BYZ_STAMP(3);   // sweep A done
    if (flags[0] != 0 || keep < 1) {   // uniform: non-finite input is the general kernel's business
        if (tid == 0) {
            redo[1 + atomicAdd(redo, 1)] = static_cast<int32_t>(tile);
            atomicAdd(&g_lean_reasons[0], 1u);
        }
        return;
    }

    // ---- scan: exclusive prefix sums over the buckets, in place, both columns of a pair in one add.  Lane l of a column
    // pair's group takes the 4-bucket chunks l, l + GRP, l + 2 GRP, ...: consecutive lanes read consecutive 16 bytes, no bank
    // conflicts (a lane that owns CONSECUTIVE buckets reads at a stride of 128 bytes: every lane on one bank -- the first
    // version's scan took a quarter of the workgroup's lifetime and held the LDS pipe against the other workgroups).
    if (wave < SCAN_WAVES) {
        const int g = lane / GRP, gl = lane % GRP;
        uint32_t* const h = un + (wave * CPS + g) * PS;
        constexpr int KCH = B / (4 * GRP);   // chunks per lane
        uint32_t base = 0;                   // values in the rows of chunks before row k
#pragma unroll
        for (int k = 0; k < KCH; ++k) {
            uint4* const at = reinterpret_cast<uint4*>(h + 4 * (k * GRP + gl));
            const uint4 w4 = *at;
            const uint32_t mine = (w4.x + w4.y) + (w4.z + w4.w);
            const uint32_t upto = group_scan<GRP>(mine);
            uint32_t row_total;
            if constexpr (GRP == 64) {
                row_total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(upto), 63));
            } else {
                const uint32_t t0 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(upto), 31));
                const uint32_t t1 = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(upto), 63));
                row_total = g ? t1 : t0;
            }
            const uint32_t p0 = base + upto - mine, p1 = p0 + w4.x, p2 = p1 + w4.y, p3 = p2 + w4.z;
            *at = make_uint4(p0, p1, p2, p3);
            base += row_total;
        }
        if (gl == 0) h[B] = base;   // entry B: every live value of the two columns (the searches clamp their index to 0 .. B)
    }
    __syncthreads();
    BYZ_STAMP(4);   // scan done

    // ---- search: 16 lanes per column (waves 0 .. 3, four columns each) on the prefix sums.  Every step of a search
    // probes 16 positions at once, so the bucket of a rank or the ring j* takes three dependent LDS round trips where a
    // binary search by one lane took ten (measured: a quarter of the workgroup's lifetime went into that chain).
    if (wave < 4) {
        const int c = 4 * wave + (lane >> 4), t = lane & 15;
        const uint32_t* const pre = un + (c >> 1) * PS;
        const int shift = 16 * (c & 1);
        auto cum = [&](int b) __attribute__((always_inline)) -> int {   // values in buckets < b: entry 0 is 0, entry B the total
            const int bb = min(max(b, 0), B);   // (one v_med3_i32)
            return static_cast<int>((pre[bb] >> shift) & 0xffffu);
        };
        auto trues = [&](bool pred) __attribute__((always_inline)) -> int {   // how many lanes of this column's group say yes
            const unsigned long long m = __ballot(pred);
            return __popc(static_cast<uint32_t>(m >> (lane & 48)) & 0xffffu);
        };
        constexpr int S1 = B / 16, S2 = S1 / 16;   // strides of the first two levels; the third probes S2 neighbours
        const int r1 = (n_rows - 1) >> 1, r2 = n_rows >> 1;
        // bucket of rank r = the last bucket with at most r values before it (cum(0) = 0: position 0 always qualifies)
        int bm1 = 0, bm2 = 0;
        {
            const int n1 = trues(cum(S1 * t) <= r1), n2 = trues(cum(S1 * t) <= r2);
            bm1 = S1 * (n1 - 1);
            bm2 = S1 * (n2 - 1);
        }
        {
            const int n1 = trues(cum(bm1 + S2 * t) <= r1), n2 = trues(cum(bm2 + S2 * t) <= r2);
            bm1 += S2 * (n1 - 1);
            bm2 += S2 * (n2 - 1);
        }
        if constexpr (S2 > 1) {
            const int n1 = trues(t < S2 && cum(bm1 + t) <= r1), n2 = trues(t < S2 && cum(bm2 + t) <= r2);
            bm1 += n1 - 1;
            bm2 += n2 - 1;
        }
        // j* = the first ring j with N(j) = values in buckets bm1 - j .. bm2 + j >= keep = how many j in 0 .. B - 1 have
        // N(j) < keep (N is monotone; N(B) = n_rows >= keep)
        auto short_of = [&](int j) __attribute__((always_inline)) -> bool { return cum(bm2 + j + 1) - cum(bm1 - j) < keep; };
        int j_star = 0;   // invariant: every j < j_star is short
        j_star += S1 * trues(short_of(j_star + S1 * (t + 1) - 1));
        j_star += S2 * trues(short_of(j_star + S2 * (t + 1) - 1));
        if constexpr (S2 > 1) j_star += trues(t < S2 - 1 && short_of(j_star + t));
        const int s = 1 + bm2 - bm1;
        const int j_in = j_star - 1 - s;               // rings <= j_in are decided in
        const int ring_hi = j_star + s + 1;            // rings beyond are decided out, a whole bucket clear of the edge
        const int ring_lo = j_in + 1;                  // may be <= 0: then the median buckets are undecided too
        const int first_left = bm1 - ring_hi;          // leftmost gathered bucket (may be < 0)
        const int n_upto_hi = cum(bm2 + ring_hi + 1) - cum(first_left);
        const int n_in = j_in >= 0 ? cum(bm2 + j_in + 1) - cum(bm1 - j_in) : 0;
        const int middle = cum(bm2 + 1) - cum(bm1);
        const int left = cum(bm1 - (ring_lo > 1 ? ring_lo : 1) + 1) - cum(first_left);
        const int at_bm1 = cum(bm1);
        const int marked = ring_lo >= 1 ? middle : 0;  // gathered for the median only, not undecided
        const int expected = n_upto_hi - n_in + marked;
        const int need = keep - n_in;
        float lo_c, hi_c, inv_c, kadd_c;
        column_range(c, lo_c, hi_c);
        column_scale(lo_c, hi_c, inv_c, kadd_c);
        // a bucket must be many ulps of the values wide (differences x - med then order like the buckets do), and the
        // addend must keep its integer part: both follow from |value| * inv < 2^17
        const float big = __builtin_fmaxf(__builtin_fabsf(lo_c), __builtin_fabsf(hi_c)) * inv_c;
        const bool range_ok = is_finite(lo_c) && is_finite(hi_c) && hi_c - lo_c > 0.0f && is_finite(inv_c) && big < 131072.0f;
        bool ok = range_ok && bm2 - bm1 <= 8 && expected <= CAP && need >= 1 && need <= n_upto_hi - n_in;
        // a sampled range: the end buckets also hold whatever fell outside it -- no ring may touch them
        if constexpr (!EXACT) ok = ok && first_left >= 1 && bm2 + ring_hi <= B - 2;
        if (t == 0) {
            if (!ok) {
                const int why = !range_ok ? 1 : bm2 - bm1 > 8 ? 2 : expected > CAP ? 3
                                : (need < 1 || need > n_upto_hi - n_in) ? 4 : 5;
                atomicAdd(&g_lean_reasons[why], 1u);
                flags[1] = 1;
            }
            ColumnPlan p;
            p.sum_b = static_cast<float>(bm1 + bm2);
            p.mid = static_cast<float>(ring_lo + s - 1);
            p.half = ok && ring_lo >= 2 ? static_cast<float>(ring_lo) : -1.0f;
            p.outer = ok ? static_cast<float>(2 * ring_hi + s - 1) : -1.0f;
            p.pivot = ok ? (kMagic + static_cast<float>(bm1) - kadd_c) / inv_c : 0.0f;     // u = bm1  <=>  x = (bm1 - nlo) / inv
            p.expected = expected;
            p.need = need;
            p.idx1 = left + (r1 - at_bm1);
            p.idx2 = left + (r2 - at_bm1);
            p.marked = marked;
            p.mid_lo = left;
            plan[c] = p;
        }
    }
    __syncthreads();   // plans written; the prefix sums are dead: their memory becomes the gather stacks
    BYZ_STAMP(5);   // search done
    if (flags[1] != 0) {   // uniform
        if (tid == 0) redo[1 + atomicAdd(redo, 1)] = static_cast<int32_t>(tile);
        return;
    }

    // ---- sweep B: gather (median buckets and undecided rings go on the lane's stack) and sum (decided-in rings)
    {
        const f32x2 magic2 = {kMagic, kMagic}, two2 = {2.0f, 2.0f};
        bool overflow = false;
#pragma unroll
        for (int ep = 0; ep < 2; ++ep) {   // the lane's columns two at a time: the affine steps run as packed instructions
            const ColumnPlan p0 = plan[4 * q + 2 * ep], p1 = plan[4 * q + 2 * ep + 1];
            f32x2 inv2, k2;
            {
                float lo, hi, iv, ka;
                column_range(4 * q + 2 * ep, lo, hi);
                column_scale(lo, hi, iv, ka);
                inv2.x = iv;
                k2.x = ka;
                column_range(4 * q + 2 * ep + 1, lo, hi);
                column_scale(lo, hi, iv, ka);
                inv2.y = iv;
                k2.y = ka;
            }
            const f32x2 nsum2 = {-p0.sum_b, -p1.sum_b}, piv2 = {p0.pivot, p1.pivot};
            uint32_t* const mine0 = un + (2 * ep) * (LS + 1) * T + tid;
            uint32_t* const mine1 = un + (2 * ep + 1) * (LS + 1) * T + tid;
            int top0 = 0, top1 = 0;
            float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const f32x2 v = {x[j][2 * ep], x[j][2 * ep + 1]};
                const f32x2 y = __builtin_elementwise_fma(v, inv2, k2);             // 2^23 + bucket
                const f32x2 z = __builtin_elementwise_fma(y - magic2, two2, nsum2);   // 2 b - (bm1 + bm2); NaN / inf: in no class
                const f32x2 d = v - piv2;
                const float az0 = __builtin_fabsf(z.x), az1 = __builtin_fabsf(z.y);
                const lane_mask in0 = lanes_lt(__builtin_fabsf(az0 - p0.mid), p0.half), in1 = lanes_lt(__builtin_fabsf(az1 - p1.mid), p1.half);
                const lane_mask hit0 = lanes_le(az0, p0.outer) & ~in0, hit1 = lanes_le(az1, p1.outer) & ~in1;
                acc0 = __fadd_rn(acc0, keep_mask(d.x, in0));
                acc1 = __fadd_rn(acc1, keep_mask(d.y, in1));
                mine0[min(top0, LS) * T] = __float_as_uint(v.x);
                mine1[min(top1, LS) * T] = __float_as_uint(v.y);
                top0 = add_mask(top0, hit0);
                top1 = add_mask(top1, hit1);
            }
            // the pair's counts and sums leave the registers at once (the tile fills half the register file).  One partial
            // per wave and column, in a fixed order: inside the 16-lane row, then across the four rows
            overflow = overflow || top0 > LS || top1 > LS;
            reinterpret_cast<uint8_t*>(tops)[4 * tid + 2 * ep] = static_cast<uint8_t>(top0 > LS ? LS : top0);
            reinterpret_cast<uint8_t*>(tops)[4 * tid + 2 * ep + 1] = static_cast<uint8_t>(top1 > LS ? LS : top1);
            acc0 = row_quad_sum(acc0);
            acc0 = __fadd_rn(acc0, lane_xor(acc0, 16, lane));
            acc0 = __fadd_rn(acc0, lane_xor(acc0, 32, lane));
            acc1 = row_quad_sum(acc1);
            acc1 = __fadd_rn(acc1, lane_xor(acc1, 16, lane));
            acc1 = __fadd_rn(acc1, lane_xor(acc1, 32, lane));
            if (rr == 0) {
                part_sum[wave * kTileCols + 4 * q + 2 * ep] = acc0;
                part_sum[wave * kTileCols + 4 * q + 2 * ep + 1] = acc1;
            }
        }
        if (overflow) {
            flags[1] = 1;
            atomicAdd(&g_lean_reasons[6], 1u);
        }
    }
    __syncthreads();
    BYZ_STAMP(6);   // sweep B done

    // ---- owners: wave w resolves columns NCW w .. NCW w + NCW - 1: compact the stacks of the column's 16 W lanes, sort,
    // median, merge, threshold, and the sum over the kept gathered values
    if (flags[1] == 0) {
        constexpr int SRC = (16 * W + 63) / 64;   // source lanes per owner lane
        float sv[NCW][SR];
        int total[NCW];
        bool all_ok = true;
#pragma unroll
        for (int k = 0; k < NCW; ++k) {
            const int c = NCW * wave + k;
            const int cq = c >> 2, ce = c & 3;
            int cnt[SRC], mine_total = 0;
#pragma unroll
            for (int sidx = 0; sidx < SRC; ++sidx) {
                const int src = lane * SRC + sidx;             // source number 0 .. 16 W - 1 -> thread 4 src + cq
                cnt[sidx] = src < 16 * W ? static_cast<int>((tops[4 * src + cq] >> (8 * ce)) & 0xffu) : 0;
                mine_total += cnt[sidx];
            }
            const int scan = static_cast<int>(group_scan<64>(static_cast<uint32_t>(mine_total)));
            total[k] = __builtin_amdgcn_readlane(scan, 63);
            int at = scan - mine_total;
            const bool ok = total[k] == sgpr(plan[c].expected) && total[k] <= CAP;   // the histogram and the gather must agree
            all_ok = all_ok && ok;
            if (ok) {
#pragma unroll
                for (int sidx = 0; sidx < SRC; ++sidx) {
                    const int src = lane * SRC + sidx;
                    const uint32_t* stack = un + ce * (LS + 1) * T + (4 * src + cq);
                    // a lane's stack rarely holds more than one or two values: the deeper slots sit behind a uniform test
#pragma unroll
                    for (int sl = 0; sl < 2; ++sl)
                        if (sl < cnt[sidx]) dense[c * CAP + at + sl] = __uint_as_float(stack[sl * T]);
                    if (__ballot(cnt[sidx] > 2) != 0ull) {
#pragma unroll
                        for (int sl = 2; sl < LS; ++sl)
                            if (sl < cnt[sidx]) dense[c * CAP + at + sl] = __uint_as_float(stack[sl * T]);
                    }
                    at += cnt[sidx];
                }
            }
        }
        if (__ballot(!all_ok) != 0ull) {
            if (lane == 0) {
                flags[1] = 1;
                atomicAdd(&g_lean_reasons[7], 1u);
            }
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < NCW; ++k)
#pragma unroll
                for (int r = 0; r < SR; ++r)
                    sv[k][r] = (r + SR * lane) < total[k] ? dense[(NCW * wave + k) * CAP + r + SR * lane] : pinf;
            wave_bitonic_sort<SR, NCW>(sv, lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < NCW; ++k)
#pragma unroll
                for (int r = 0; r < SR; ++r) dense[(NCW * wave + k) * CAP + r + SR * lane] = sv[k][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool even = (n_rows & 1) == 0;
            float med[NCW], sx[NCW][SR];
            int marked[NCW], mid_lo[NCW], need[NCW];
#pragma unroll
            for (int k = 0; k < NCW; ++k) {
                const int c = NCW * wave + k;
                marked[k] = sgpr(plan[c].marked);
                mid_lo[k] = sgpr(plan[c].mid_lo);
                need[k] = sgpr(plan[c].need);
                const float a = dense[c * CAP + sgpr(plan[c].idx1)], b = dense[c * CAP + sgpr(plan[c].idx2)];
                med[k] = sgpr(even ? __fmul_rn(__fadd_rn(a, b), 0.5f) : a);    // np.median
                // deviations of the sorted gathered values: falling, then rising; the median buckets that are not
                // undecided sink to the bottom (-inf) and are skipped by rank, the padding floats on top (+inf)
#pragma unroll
                for (int r = 0; r < SR; ++r) {
                    const int i = r + SR * lane;
                    sx[k][r] = sv[k][r];
                    const float dv = __builtin_fabsf(__fsub_rn(sv[k][r], med[k]));
                    const bool skip = marked[k] > 0 && i >= mid_lo[k] && i < mid_lo[k] + marked[k];
                    sv[k][r] = i >= total[k] ? pinf : (skip ? -pinf : dv);
                }
            }
            wave_bitonic_merge<SR, NCW>(sv, lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < NCW; ++k)
#pragma unroll
                for (int r = 0; r < SR; ++r) dense[(NCW * wave + k) * CAP + r + SR * lane] = sv[k][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < NCW; ++k) {
                const int c = NCW * wave + k;
                const float thr = sgpr(dense[c * CAP + marked[k] + need[k] - 1]);
                const float pivot = sgpr(plan[c].pivot);
                // the kept gathered values: the marked median buckets and the undecided values within the threshold --
                // exactly `need` of the latter, or values tie at the window's edge (row order decides: the general kernel)
                float part = 0.0f;
                int within = 0;
#pragma unroll
                for (int r = 0; r < SR; ++r) {
                    const int i = r + SR * lane;
                    const bool skip = marked[k] > 0 && i >= mid_lo[k] && i < mid_lo[k] + marked[k];
                    const bool und = i < total[k] && !skip && __builtin_fabsf(__fsub_rn(sx[k][r], med[k])) <= thr;
                    within += und ? 1 : 0;
                    part = __fadd_rn(part, (und || (skip && i < total[k])) ? __fsub_rn(sx[k][r], pivot) : 0.0f);
                }
                const int n_within = wave_sum_i(within);
                float sum = wave_sum_f(part, lane);
                // ... plus the decided-in rings, summed by sweep B: the partials in a fixed order
                for (int p = 0; p < W; ++p) sum = __fadd_rn(sum, part_sum[p * kTileCols + c]);
                if (lane == 0) {
                    if (n_within != need[k] || !(thr == thr)) {
                        flags[1] = 1;
                        atomicAdd(&g_lean_reasons[thr == thr ? 8 : 9], 1u);
                    }
                    colres[c] = __fadd_rn(__fdiv_rn(sum, static_cast<float>(keep)), pivot);   // defences.py:51
                }
            }
        }
    }
    __syncthreads();
    BYZ_STAMP(7);   // owners done
    if (flags[1] != 0) {
        if (tid == 0) redo[1 + atomicAdd(redo, 1)] = static_cast<int32_t>(tile);
    } else if (tid < kTileCols && c_base + tid < n_cols) {
        out[c_base + tid] = colres[tid];
    }
#undef BYZ_STAMP
}

template <int W, int RPW, int B, int SR, int LS>
int launch_lean_shape(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index, int64_t keep,
                      float* out, int32_t* redo, hipStream_t stream) {
    const int64_t n_tiles = ceil_div(n_cols, static_cast<int64_t>(kTileCols));
    constexpr int kHist = 8 * (B + 4), kStack = 4 * (LS + 1) * 64 * W;
    constexpr size_t lds = static_cast<size_t>(kHist > kStack ? kHist : kStack) * sizeof(uint32_t);
    // the sampled range resolves a column only while its window stays clear of the end buckets: columns that keep nearly
    // all of their rows (and short columns, where the sample is most of the column anyway) sweep for the true range
    const bool exact = keep * 10 > n_rows * 9 || n_rows < 192;
    const char* timing_env = std::getenv("BYZ_TM_LEAN_TIMING");
    // (tiles from the middle of the launch: the chip is in its steady state there)
    const int timing = timing_env != nullptr && std::atoi(timing_env) != 0 ? 1 + static_cast<int>(n_tiles > 2 * kStampTiles ? n_tiles / 2 : 0) : 0;
    const char* xcd_env = std::getenv("BYZ_TM_LEAN_XCD");   // 0: tile = blockIdx.x (the comparison)
    const int by_xcd = xcd_env != nullptr ? std::atoi(xcd_env) : 1;
    // PIPE: request the tile's rows under the range phase and the histogram instead of all at once.  Same box, tiles ordered by
    // XCD (scripts/tm_ab.py BYZ_TM_LEAN_PIPE 0,1): 2080 rows 0.398 -> 0.376 ms per 2^17 columns, 5200 rows 0.675 -> 0.626 per
    // 2^16, 1000 rows 0.351 -> 0.345 per 2^18 (inside the noise: the 4-wave shapes keep loading at once).  Before the tile
    // order it paid for the 16-wave shapes only.  BYZ_TM_LEAN_PIPE=0|1 forces it either way: the comparison.
    const char* pipe_env = std::getenv("BYZ_TM_LEAN_PIPE");
    const bool pipe = pipe_env != nullptr ? std::atoi(pipe_env) != 0 : W >= 8;
#define BYZ_LEAN(E, P)                                                                                              \
    do {                                                                                                            \
        BYZ_HIP(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&window_lean_kernel<W, RPW, B, SR, LS, E, P>),  \
                                  static_cast<int>(lds)));                                                          \
        window_lean_kernel<W, RPW, B, SR, LS, E, P><<<static_cast<unsigned>(n_tiles), 64 * W, lds, stream>>>(        \
            G, static_cast<int>(n_rows), n_cols, ld, row_index, static_cast<int>(keep), out, redo, by_xcd, timing); \
    } while (0)
    if (exact) BYZ_LEAN(true, false);
    else if (pipe) BYZ_LEAN(false, true);
    else BYZ_LEAN(false, false);
#undef BYZ_LEAN
    BYZ_TRY(check_launch("window_lean_kernel"));
    if (timing) {
        static unsigned long long stamps[kStampTiles * kStamps];
        BYZ_HIP(hipStreamSynchronize(stream));
        BYZ_HIP(hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_lean_stamps), sizeof(stamps)));
        const int tiles = n_tiles < kStampTiles ? static_cast<int>(n_tiles) : kStampTiles;
        double sum[kStamps] = {0};
        for (int t = 0; t < tiles; ++t)
            for (int i = 1; i < 8; ++i) sum[i] += static_cast<double>(stamps[t * kStamps + i] - stamps[t * kStamps + i - 1]);
        {
            double issue_only = 0;
            for (int t = 0; t < tiles; ++t) issue_only += static_cast<double>(stamps[t * kStamps + 8] - stamps[t * kStamps + 0]);
            std::fprintf(stderr, "lean: the load instructions alone take %.0f ticks of the first phase\n", issue_only / tiles);
        }
        std::fprintf(stderr, "lean W=%d RPW=%d phases (s_memtime ticks, mean of %d tiles): issue %.0f | load wait + range %.0f | sweep A %.0f | scan %.0f | search %.0f | sweep B %.0f | owners %.0f\n",
                     W, RPW, tiles, sum[1] / tiles, sum[2] / tiles, sum[3] / tiles, sum[4] / tiles, sum[5] / tiles, sum[6] / tiles, sum[7] / tiles);
    }
    if (const char* dbg = std::getenv("BYZ_TM_LEAN_DEBUG"); dbg != nullptr && std::atoi(dbg) != 0) {
        unsigned int host[16] = {0};
        BYZ_HIP(hipStreamSynchronize(stream));
        BYZ_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lean_reasons), sizeof(host)));
        std::fprintf(stderr, "lean W=%d RPW=%d %s: %lld tiles; not resolved because of", W, RPW, exact ? "exact" : "sampled", (long long)n_tiles);
        for (int i = 0; i < 10; ++i) std::fprintf(stderr, " [%d] %u", i, host[i]);
        std::fprintf(stderr, "\n");
        const unsigned int zero[16] = {0};
        BYZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_lean_reasons), zero, sizeof(zero)));
    }
    return BYZ_OK;
}

}  // namespace

// The lean ring selection over row-split tiles; unresolved tiles are appended to redo[1 ...] (redo[0] counts them).
// Returns BYZ_E_UNSUPPORTED for a height it has no instantiation for (the caller keeps its other kernels).
int launch_window_lean(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                       int64_t keep, float* out, int32_t* redo, hipStream_t stream) {
    if (ld >= (int64_t{1} << 30)) return BYZ_E_UNSUPPORTED;   // the kernel forms row offsets as 32 x 32 -> 64-bit products
    if (n_rows < 65 || keep < 1) return BYZ_E_UNSUPPORTED;
    const int64_t blocks = ceil_div(n_rows, 16);
#define BYZ_SHAPE(W, RPW, B, SR, LS) \
    if (blocks <= (W) * (RPW)) return launch_lean_shape<W, RPW, B, SR, LS>(ctx, G, n_rows, n_cols, ld, row_index, keep, out, redo, stream)
    // (Twice the waves per tile with half the rows per lane -- (8, 8) for 1000 rows: 76 registers, three workgroups per CU;
    // (16, 9) for 2080 rows: 64 registers with spills -- measured slower: 0.445 vs 0.420 ms and 1.67 vs 0.95 ms per 2^18
    // columns.  Not kept.)
    // Bucket counts: about one value per bucket at the column's centre.  The candidates of a column are the values of
    // 4 s + 5 buckets (s = 1, 2); with half as many buckets 0.3% (1000 rows) to 3% (5200 rows) of the columns exceeded the
    // sort's capacity and their tiles went to the general kernel (profiles/r03e: 5% .. 43% of the tiles).
    BYZ_SHAPE(4, 4, 512, 1, 6);     //  <=  256 rows
    BYZ_SHAPE(4, 8, 512, 1, 6);     //  <=  512
    BYZ_SHAPE(4, 12, 1024, 1, 6);   //  <=  768
    BYZ_SHAPE(4, 16, 1024, 1, 6);   //  <= 1024
    BYZ_SHAPE(8, 12, 2048, 2, 6);   //  <= 1536
    BYZ_SHAPE(8, 17, 2048, 2, 6);   //  <= 2176
    BYZ_SHAPE(8, 20, 2048, 2, 6);   //  <= 2560
    BYZ_SHAPE(16, 14, 2048, 4, 6);  //  <= 3584
    BYZ_SHAPE(16, 21, 2048, 4, 6);  //  <= 5376
#undef BYZ_SHAPE
    return BYZ_E_UNSUPPORTED;
}

}  // namespace byz
