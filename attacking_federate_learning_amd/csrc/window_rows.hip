// Median-window trimmed mean (reference defences.py:44-52), row-split layout: the fast path for 65 .. 5376 rows.
//
// median_window.hip gives every wave four whole columns, which forces every value through an LDS transposition (global
// loads are row segments) with two workgroup barriers per 256 rows.  Here the WORKGROUP owns the 16 columns of a tile and
// the waves split the ROWS: one global load instruction of a wave is 16 rows x 64 bytes, lane l = (row l >> 2, column
// quad l & 3), and it lands where the value stays -- no transposition, every load of the tile in flight at once, six
// barriers per tile whatever its height.  What used to be per-wave state becomes workgroup state in LDS:
//
//   sweep 1  column min / max                      DPP inside a row of lanes, LDS atomics across rows and waves
//   sweep 2  one histogram per column over B equal-width buckets of [min, max] (bucket index monotone in the value)
//   owners   16 / W columns per wave: bucket prefix sums; the median bucket(s) bm1, bm2; the ring j* that completes
//            `keep` values around them (ring(b) = max(bm1 - b, b - bm2, 0)); rings j* - s .. j* + s (s = 1 + bm2 - bm1)
//            are undecided, everything inside them belongs to the window, everything outside does not
//   sweep 3  gather of the median buckets and the undecided rings (a few dozen values per column) into per-lane stacks
//   owners   one 64-lane sort by value: the median exactly as np.median forms it; |fl(x - med)| over sorted values falls
//            and then rises, so one bitonic merge yields T = the (keep - N_in)-th smallest undecided deviation
//   sweep 4  sum of fl(x - med) over |.| <= T, in a fixed order; exactly `keep` values must pass
//
// Whatever this cannot resolve (ties at the window edge, more candidates than the sort takes, non-finite input, a
// degenerate range) goes, tile by tile, to the general kernel of median_window.hip through the redo list, as before.
// scripts/proto/ring_window.py is the selection in numpy, bit for bit the oracle on every column it resolves.
//
// ~25 vector instructions per value (min/max 3, histogram 5, gather 12, sum 6), no staging.  Bound: HBM, 4 R D + 4 D bytes.
#include "common.hpp"

#include "lane_exchange.hpp"

#include <cstdlib>

namespace byz {
namespace {

using namespace lanes;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kTileCols = 16;

__device__ __forceinline__ uint32_t okey(float v) {   // order-preserving float -> uint32
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_okey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ bool is_finite(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float sgpr(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

__device__ __forceinline__ int wave_sum_i(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);   // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
           __builtin_amdgcn_readlane(x, 48);
}
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
__device__ __forceinline__ int row_quad_sum_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);
    return v;
}

__device__ __forceinline__ int hslot(int b) { return b + (b >> 3); }

struct ColumnPlan {   // what sweep 3 needs to know about a column (written by its owner)
    int bm1, bm2, ring_lo, span;
};

// W waves, RPW 16-row blocks per wave (rows <= 16 W RPW), B buckets, SR sort registers per lane (64 SR candidates),
// LS stack slots per lane and column.  H16: 16-bit histogram counters (rows < 65536), two per LDS word -- the 8-wave shapes'
// LDS then is what their gather stacks need, and TWO workgroups fit a CU (the default: see launch_window_rows).
template <int W, int RPW, int B, int SR, int LS, bool H16>
__global__ __launch_bounds__(64 * W, (W == 4 ? (H16 ? 4 : 3) : 4)) void window_rows_kernel(const float* __restrict__ G, int n_rows, int64_t n_cols,
                                                             int64_t ld, const int32_t* __restrict__ row_index, int keep,
                                                             float* __restrict__ out, int32_t* __restrict__ redo) {
    constexpr int T = 64 * W;
    constexpr int NCW = kTileCols / W;          // columns an owner wave resolves: 4, 2 or 1
    constexpr int BPL = B / 64;                 // buckets per lane in the owners' scans
    constexpr int CAP = 64 * SR;
    constexpr int kColWords = B + B / 8;          // a column's histogram, skewed: bucket b sits at slot b + b / 8
    constexpr int kHistWords = H16 ? kColWords * kTileCols / 2 : kColWords * kTileCols;   // 32-bit words
    extern __shared__ __attribute__((aligned(16))) uint32_t un[];       // max(B x 16, 4 (LS + 1) T) words: the histogram, then the gather stacks
    auto hist_get = [&](int slot) __attribute__((always_inline)) -> int {
        if constexpr (H16) return static_cast<int>(reinterpret_cast<const uint16_t*>(un)[slot]);
        else return static_cast<int>(un[slot]);
    };
    auto hist_set = [&](int slot, int v) __attribute__((always_inline)) {
        if constexpr (H16) reinterpret_cast<uint16_t*>(un)[slot] = static_cast<uint16_t>(v);
        else un[slot] = static_cast<uint32_t>(v);
    };
    __shared__ float dense[kTileCols * CAP];
    __shared__ uint32_t tops[T];
    __shared__ uint32_t minmax[2 * kTileCols];
    __shared__ ColumnPlan plan[kTileCols];
    __shared__ float medthr[2 * kTileCols];
    __shared__ float part_sum[W * 4 * kTileCols];
    __shared__ int part_cnt[W * 4 * kTileCols];
    __shared__ int flags[2];   // [0] non-finite input seen, [1] a column could not be resolved

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = sgpr(tid >> 6);
    const int rr = lane >> 2, q = lane & 3;
    const int64_t tile = blockIdx.x;
    const int64_t c_base = tile * kTileCols;
    const float pinf = __builtin_inff();

    // ---- every load of the tile, at once.  The loads carry no branch (a branch around a load makes hipcc wait for it
    // before the next one): rows past the matrix re-read the last row and are overwritten with the padding afterwards.
    // Addresses are one 32 x 32 -> 64-bit multiply-add per load (ld * 4 < 2^32 is the launcher's precondition).
    f32x4 x[RPW];
    {
        const int64_t col = c_base + 4 * q;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(G) + col * 4;
        const uint32_t pitch = static_cast<uint32_t>(ld) * 4u;
        if (c_base + kTileCols <= n_cols) {   // uniform: every tile but a ragged last one
            uint32_t src[RPW];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                int row = (j * W + wave) * 16 + rr;
                row = row < n_rows ? row : n_rows - 1;
                src[j] = static_cast<uint32_t>(row_index ? row_index[row] : row);
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j)
                x[j] = *reinterpret_cast<const f32x4u*>(base + static_cast<uint64_t>(src[j]) * pitch);
        } else {
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                int row = (j * W + wave) * 16 + rr;
                row = row < n_rows ? row : n_rows - 1;
                const uint32_t src = static_cast<uint32_t>(row_index ? row_index[row] : row);
                const float* ptr = reinterpret_cast<const float*>(base + static_cast<uint64_t>(src) * pitch);
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};   // columns past the matrix are computed on zeros and never stored
                if (col + 0 < n_cols) v.x = ptr[0];
                if (col + 1 < n_cols) v.y = ptr[1];
                if (col + 2 < n_cols) v.z = ptr[2];
                if (col + 3 < n_cols) v.w = ptr[3];
                x[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) {   // padding rows: +inf (never a minimum, masked out of everything else)
            if ((j * W + wave) * 16 + 15 >= n_rows) {   // wave-uniform: only the last block or two can hold padding
                if ((j * W + wave) * 16 + rr >= n_rows) x[j] = f32x4{pinf, pinf, pinf, pinf};
            }
        }
    }
    // LDS set-up while the loads fly
    for (int i = tid; i < kHistWords / 4; i += T) reinterpret_cast<uint4*>(un)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < kTileCols) {
        minmax[tid] = 0xffffffffu;               // min of ordered keys
        minmax[kTileCols + tid] = 0u;            // max
    }
    if (tid < 2) flags[tid] = 0;
    __syncthreads();

    // ---- sweep 1: column minimum and maximum; x * 0 turns NaN as soon as one value is NaN or +-inf
    {
        float mn[4] = {pinf, pinf, pinf, pinf}, mx[4] = {-pinf, -pinf, -pinf, -pinf};
        float poison = 0.0f;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            if ((j * W + wave) * 16 + 15 < n_rows) {   // wave-uniform: a block without padding
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mn[e] = __builtin_fminf(mn[e], x[j][e]);
                    mx[e] = __builtin_fmaxf(mx[e], x[j][e]);
                }
                poison = __builtin_fmaf(x[j][0] + x[j][1], 0.0f, __builtin_fmaf(x[j][2] + x[j][3], 0.0f, poison));
            } else {
                const bool live = (j * W + wave) * 16 + rr < n_rows;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    mn[e] = __builtin_fminf(mn[e], x[j][e]);
                    mx[e] = __builtin_fmaxf(mx[e], live ? x[j][e] : -pinf);
                }
                if (live) poison = __builtin_fmaf(x[j][0] + x[j][1], 0.0f, __builtin_fmaf(x[j][2] + x[j][3], 0.0f, poison));
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
        if (poison != poison) flags[0] = 1;
    }
    __syncthreads();
    const bool suspicious = flags[0] != 0 || keep < 1;

    // ---- sweep 2: histograms ([bucket][column], 32-bit counts)
    float inv[4], nlo[4];   // the same two numbers map a value to its bucket in sweep 2 and in sweep 3
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float lo = from_okey(minmax[4 * q + e]), hi = from_okey(minmax[kTileCols + 4 * q + e]);
        inv[e] = static_cast<float>(B) * (1.0f - 1.0f / 1048576.0f) / (hi - lo);
        nlo[e] = -lo * inv[e];
    }
    // (a column with a degenerate range stops its owner below; the others go on)
    if (!suspicious) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            if ((j * W + wave) * 16 + 15 < n_rows || (j * W + wave) * 16 + rr < n_rows) {   // (uniform test first)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (the clamp comes before the conversion: the product can exceed B by rounding)
                    const int b = static_cast<int>(__builtin_fminf(__builtin_fmaf(x[j][e], inv[e], nlo[e]), static_cast<float>(B) - 0.5f));
                    {
                        const int bb = max(b, 0);
                        const int slot = (4 * q + e) * kColWords + bb + (bb >> 3);
                        if constexpr (H16) atomicAdd(&un[slot >> 1], 1u << (16 * (slot & 1)));
                        else atomicAdd(&un[slot], 1u);
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- owners: wave w resolves columns NCW w .. NCW w + NCW - 1
    const bool even = (n_rows & 1) == 0;
    const int r1 = (n_rows - 1) >> 1, r2 = n_rows >> 1;
    int need[NCW], idx1[NCW], idx2[NCW], marked[NCW], expected[NCW], mid_lo[NCW];
    bool col_ok[NCW];
    {
#pragma unroll
        for (int k = 0; k < NCW; ++k) {
            const int c = NCW * wave + k;
            bool ok = !suspicious;
            {
                const float lo = from_okey(minmax[c]), hi = from_okey(minmax[kTileCols + c]);
                const float width = hi - lo;
                const float iv = static_cast<float>(B) * (1.0f - 1.0f / 1048576.0f) / width;
                ok = ok && is_finite(lo) && is_finite(hi) && width > 0.0f && is_finite(iv) && is_finite(-lo * iv);
            }
            // exclusive prefix sums over the buckets, in place; bucket of rank r = number of buckets whose inclusive
            // sum is <= r
            int run = 0;
#pragma unroll
            for (int i = 0; i < BPL; ++i) run += hist_get(c * kColWords + hslot(lane * BPL + i));
            int scan = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(scan, d, 64);
                if (lane >= d) scan += up;
            }
            int acc = scan - run;
            int below1 = 0, below2 = 0;
#pragma unroll
            for (int i = 0; i < BPL; ++i) {
                const int h = hist_get(c * kColWords + hslot(lane * BPL + i));
                hist_set(c * kColWords + hslot(lane * BPL + i), acc);
                acc += h;
                below1 += acc <= r1 ? 1 : 0;
                below2 += acc <= r2 ? 1 : 0;
            }
            const int bm1 = wave_sum_i(below1), bm2 = wave_sum_i(below2);
            ok = ok && bm1 < B && bm2 < B && bm2 - bm1 <= 8;   // (a sparse histogram: the two middle values sit buckets apart)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            auto cum = [&](int b) {   // values in buckets < b
                return b <= 0 ? 0 : (b >= B ? n_rows : hist_get(c * kColWords + hslot(b)));
            };
            // j* = number of rings j with N(j) < keep (N is monotone); lane l tries j = l, l + 64, ...
            int fewer = 0;
            if (ok) {
#pragma unroll 1
                for (int i = 0; i < BPL; ++i) {
                    const int j = lane + 64 * i;
                    fewer += cum(bm2 + j + 1) - cum(bm1 - j) < keep ? 1 : 0;
                }
            }
            const int j_star = wave_sum_i(fewer);
            const int s = 1 + bm2 - bm1;
            const int j_in = j_star - 1 - s;               // rings <= j_in are decided in
            const int ring_hi = j_star + s;
            const int ring_lo = j_in + 1;                  // may be <= 0: then the median buckets are undecided too
            const int first_left = bm1 - ring_hi;          // leftmost gathered bucket (may be < 0)
            int n_upto_hi = 0, n_in = 0, middle = 0, left = 0, at_bm1 = 0;
            if (ok) {
                n_upto_hi = sgpr(cum(bm2 + ring_hi + 1) - cum(first_left));
                n_in = j_in >= 0 ? sgpr(cum(bm2 + j_in + 1) - cum(bm1 - j_in)) : 0;
                middle = sgpr(cum(bm2 + 1) - cum(bm1));
                left = sgpr(cum(bm1 - (ring_lo > 1 ? ring_lo : 1) + 1) - cum(first_left));
                at_bm1 = sgpr(cum(bm1));
            }
            marked[k] = ring_lo >= 1 ? middle : 0;         // gathered for the median only, not undecided
            expected[k] = n_upto_hi - n_in + marked[k];
            need[k] = keep - n_in;
            mid_lo[k] = left;                              // first gathered value of the median buckets
            idx1[k] = left + (r1 - at_bm1);
            idx2[k] = left + (r2 - at_bm1);
            ok = ok && expected[k] <= CAP && need[k] >= 1 && need[k] <= n_upto_hi - n_in;
            col_ok[k] = ok;
            if (lane == 0) {
                plan[c].bm1 = bm1;
                plan[c].bm2 = bm2;
                plan[c].ring_lo = ok ? ring_lo : (1 << 28);   // an unresolved column gathers nothing
                plan[c].span = ok ? ring_hi - ring_lo : 0;
                if (!ok) {
                    plan[c].bm1 = -(1 << 28);
                    plan[c].bm2 = -(1 << 28);
                    flags[1] = 1;
                }
            }
        }
    }
    __syncthreads();   // plans written; the histogram is dead: its memory becomes the gather stacks

    // ---- sweep 3: gather.  Every lane keeps a short stack per column; a miss overwrites the scratch slot on top.
    if (flags[1] == 0) {
        int top[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const ColumnPlan p = plan[4 * q + e];
            uint32_t* mine = un + e * (LS + 1) * T + tid;
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const float a = x[j][e];
                const int b = static_cast<int>(__builtin_fminf(__builtin_fmaf(a, inv[e], nlo[e]), 1.0e9f));   // padding: far out
                const int ring = max(max(p.bm1 - b, b - p.bm2), 0);
                const bool hit = static_cast<uint32_t>(ring - p.ring_lo) <= static_cast<uint32_t>(p.span) || ring == 0;
                mine[min(top[e], LS) * T] = __float_as_uint(a);
                top[e] += hit ? 1 : 0;
            }
        }
        bool overflow = false;
        uint32_t packed = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            overflow = overflow || top[e] > LS;
            packed |= static_cast<uint32_t>(top[e] > LS ? LS : top[e]) << (8 * e);
        }
        tops[tid] = packed;
        if (overflow) flags[1] = 1;
    }
    __syncthreads();

    // ---- owners: compact the stacks of a column's 16 W lanes, sort, median, merge, threshold
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
            int scan = mine_total;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(scan, d, 64);
                if (lane >= d) scan += up;
            }
            total[k] = __builtin_amdgcn_readlane(scan, 63);
            int at = scan - mine_total;
            const bool ok = total[k] == expected[k] && total[k] <= CAP;   // the histogram and the gather must agree
            all_ok = all_ok && ok;
            if (ok) {
#pragma unroll
                for (int sidx = 0; sidx < SRC; ++sidx) {
                    const int src = lane * SRC + sidx;
                    const uint32_t* stack = un + ce * (LS + 1) * T + (4 * src + cq);
#pragma unroll
                    for (int sl = 0; sl < LS; ++sl)
                        if (sl < cnt[sidx]) dense[c * CAP + at + sl] = __uint_as_float(stack[sl * T]);
                    at += cnt[sidx];
                }
            }
        }
        if (__ballot(!all_ok) != 0ull) {
            if (lane == 0) flags[1] = 1;
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
            float med[NCW];
#pragma unroll
            for (int k = 0; k < NCW; ++k) {
                const int c = NCW * wave + k;
                const float a = dense[c * CAP + idx1[k]], b = dense[c * CAP + idx2[k]];
                med[k] = sgpr(even ? __fmul_rn(__fadd_rn(a, b), 0.5f) : a);    // np.median
                // deviations of the sorted gathered values: falling, then rising; the median buckets that are not
                // undecided sink to the bottom (-inf) and are skipped by rank, the padding floats on top (+inf)
#pragma unroll
                for (int r = 0; r < SR; ++r) {
                    const int i = r + SR * lane;
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
                const float thr = dense[c * CAP + marked[k] + need[k] - 1];
                if (lane == 0) {
                    medthr[2 * c] = med[k];
                    medthr[2 * c + 1] = thr;
                    if (!(thr == thr) || !col_ok[k]) flags[1] = 1;
                }
            }
        }
    }
    __syncthreads();

    // ---- sweep 4: the window sum, in a fixed order (register order, two row rotations, then the partials in order)
    if (flags[1] == 0) {
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int inside[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float med = medthr[2 * (4 * q + e)], thr = medthr[2 * (4 * q + e) + 1];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const float d = __fsub_rn(x[j][e], med);     // +inf padding stays +inf: never inside
                const bool in = __builtin_fabsf(d) <= thr;
                acc[e] = __fadd_rn(acc[e], in ? d : 0.0f);
                inside[e] += in ? 1 : 0;
            }
            acc[e] = row_quad_sum(acc[e]);
            inside[e] = row_quad_sum_i(inside[e]);
        }
        if ((rr & 3) == 0) {
            const int slot = (wave * 4 + (rr >> 2)) * kTileCols + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                part_sum[slot + e] = acc[e];
                part_cnt[slot + e] = inside[e];
            }
        }
    }
    __syncthreads();
    if (flags[1] == 0 && tid < kTileCols) {
        float sum = 0.0f;
        int n_inside = 0;
        for (int p = 0; p < W * 4; ++p) {
            sum = __fadd_rn(sum, part_sum[p * kTileCols + tid]);
            n_inside += part_cnt[p * kTileCols + tid];
        }
        // more than `keep` values within T: ties at the edge (row order decides) -- fewer: a decided-in value lies
        // beyond T; both are the general kernel's business
        if (n_inside != keep) flags[1] = 1;
        part_sum[tid] = __fadd_rn(__fdiv_rn(sum, static_cast<float>(keep)), medthr[2 * tid]);   // defences.py:51
    }
    __syncthreads();
    if (flags[1] != 0) {
        if (tid == 0) redo[1 + atomicAdd(redo, 1)] = static_cast<int32_t>(tile);
    } else if (tid < kTileCols && c_base + tid < n_cols) {
        out[c_base + tid] = part_sum[tid];
    }
}

}  // namespace

int64_t window_rows_max_rows() { return 16 * 16 * 21; }

template <int W, int RPW, int B, int SR, int LS, bool H16 = false>
static int launch_shape(const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index, int64_t keep,
                        float* out, int32_t* redo, hipStream_t stream) {
    const int64_t n_tiles = ceil_div(n_cols, static_cast<int64_t>(kTileCols));
    constexpr int kHist = (B + B / 8) * kTileCols / (H16 ? 2 : 1), kStack = 4 * (LS + 1) * 64 * W;
    constexpr size_t lds = static_cast<size_t>(kHist > kStack ? kHist : kStack) * sizeof(uint32_t);
    BYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&window_rows_kernel<W, RPW, B, SR, LS, H16>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    window_rows_kernel<W, RPW, B, SR, LS, H16><<<static_cast<unsigned>(n_tiles), 64 * W, lds, stream>>>(
        G, static_cast<int>(n_rows), n_cols, ld, row_index, static_cast<int>(keep), out, redo);
    return check_launch("window_rows_kernel");
}

// The ring selection over row-split tiles; unresolved tiles are appended to redo[1 ...] (redo[0] counts them).
// Returns BYZ_E_UNSUPPORTED for a height it has no instantiation for (the caller keeps the column-split kernels).
int launch_window_rows(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                       int64_t keep, float* out, int32_t* redo, hipStream_t stream) {
    (void)ctx;
    if (ld >= (int64_t{1} << 30)) return BYZ_E_UNSUPPORTED;   // the kernel forms row offsets as 32 x 32 -> 64-bit products
    const int64_t blocks = ceil_div(n_rows, 16);
#define BYZ_SHAPE(W, RPW, B, SR, LS) \
    if (blocks <= (W) * (RPW)) return launch_shape<W, RPW, B, SR, LS>(G, n_rows, n_cols, ld, row_index, keep, out, redo, stream)
    // 16-bit histogram counters (rows < 65536, two per LDS word) wherever the histogram, not the gather stacks, sets the
    // workgroup's LDS: the 8-wave shapes then need 57 + 14 KiB instead of 74 + 14 and TWO workgroups fit a CU -- one loads while
    // the other selects.  Measured in round 3 (profiles/r03a_optin_variants_probe.txt): 2080 rows x 2^18 columns 1.916 -> 1.181 ms
    // (1.14 -> 1.85 TB/s), results bitwise equal, the same tiles redone.  BYZ_TM_HIST16=0 keeps 32-bit counters.
    // (Halving the bucket count instead does not work: scripts/proto/ring_window.py puts 6% of the columns of a 2080-row,
    // keep-159 tile over the 128-candidate sort with 512 buckets, i.e. most tiles.)
    const char* hist16_env = std::getenv("BYZ_TM_HIST16");
    const bool hist16 = hist16_env == nullptr || std::atoi(hist16_env) != 0;
#define BYZ_SHAPE16(W, RPW, B, SR, LS)                                                                                   \
    if (blocks <= (W) * (RPW))                                                                                           \
        return hist16 ? launch_shape<W, RPW, B, SR, LS, true>(G, n_rows, n_cols, ld, row_index, keep, out, redo, stream) \
                      : launch_shape<W, RPW, B, SR, LS>(G, n_rows, n_cols, ld, row_index, keep, out, redo, stream)
    BYZ_SHAPE16(4, 4, 512, 1, 6);     //  <=  256 rows
    BYZ_SHAPE16(4, 8, 512, 1, 6);     //  <=  512
    BYZ_SHAPE16(4, 12, 512, 1, 6);    //  <=  768
    BYZ_SHAPE16(4, 16, 512, 1, 6);    //  <= 1024
    BYZ_SHAPE16(8, 12, 1024, 2, 6);   //  <= 1536
    BYZ_SHAPE16(8, 17, 1024, 2, 6);   //  <= 2176
    BYZ_SHAPE16(8, 20, 1024, 2, 6);   //  <= 2560
    BYZ_SHAPE(16, 14, 1024, 4, 6);    //  <= 3584   (16 waves: the gather stacks set the LDS, 16-bit counters gain nothing)
    BYZ_SHAPE(16, 21, 1024, 4, 6);    //  <= 5376
#undef BYZ_SHAPE16
#undef BYZ_SHAPE
    return BYZ_E_UNSUPPORTED;
}

}  // namespace byz
