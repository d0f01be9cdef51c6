// Median-window trimmed mean, reference defences.py:44-52, per parameter (column) over the client rows.
//
//   med  = np.median(column)                       fp32; even count -> (a + b) / 2
//   good = sorted(column - med, key=abs)[:k]       stable: ties in |x - med| keep the lower row first
//   out  = np.mean(good) + med
//
// This is the HBM-bound kernel of the path: 4 bytes read per (client, parameter), 4 bytes written per
// parameter, and a sort per parameter that must hide under the load.
//
// Register-resident kernel (rows <= 1024), per workgroup of 8 waves:
//   * tile = all rows x 32 consecutive parameters, loaded as 128-byte row segments (8 lanes x dwordx4)
//     into LDS with a 36-float row stride (16-byte aligned for b128 traffic, and a b128 read of 16
//     consecutive rows at one column offset touches 16 different 16-byte slots: conflict-free);
//   * each wave takes 4 adjacent columns (one float4 per row): lane l reads rows l, l+64, ... with
//     ds_read_b128, so a lane holds R = n_pad/64 values of each of its 4 columns in registers;
//   * the 64*R values of a column are sorted by a bitonic network in the "flip" form (every
//     compare-exchange puts the smaller value at the lower index).  With element index i = r + R*lane the
//     low log2(R) index bits are register bits: those steps are plain v_min/v_max pairs.  Steps on lane
//     bits fetch the partner through DPP (quad_perm / row_mirror / row_half_mirror / row_ror), ds_swizzle
//     or ds_bpermute and keep min or max with one v_med3_f32 against a per-lane +/-inf;
//   * the sorted columns go back to the wave's own 4 columns of the tile in rank order (skewed so the
//     b128 stores do not collide), and the window is found from the sorted data:
//       - the k kept values are the k nearest neighbours of the median, a contiguous rank window
//         [lo, lo + k).  lo = number of ranks i in [0, n-k) whose value is farther from the median than
//         rank i + k (counted with one ballot per 64 candidates);
//       - a cross-side tie |x - med| == |y - med| at the window edge is the only place where the row
//         order matters; it is detected exactly and resolved by rescanning the column in row order.
// General kernel (rows up to 8192): same post-processing, the sort runs in LDS over a [rank][4] float4
// array per workgroup.  It is the fallback for Bulyan's second stage at large theta.
#include "common.hpp"

#include "lane_exchange.hpp"

#include <cstdlib>

namespace byz {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kGeneralMaxRows = 16384;   // = the selection kernels' limit: defend['TrimmedMean'] then serves every N that Krum / Bulyan do

using namespace lanes;

// ---- post-processing on a sorted quad of columns -------------------------------------------------
// `Sorted` exposes at(rank) -> float4 (the 4 columns' values at that rank).  One wave per call.
struct WindowArgs {
    const float* G;
    int64_t ld;
    const int32_t* row_index;
    int n;      // rows
    int keep;   // number of kept values, 0 <= keep <= n
    int64_t col0;   // first of the 4 columns
    int64_t n_cols;
};

__device__ __forceinline__ float quad_get(const f32x4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

template <typename Sorted>
__device__ __forceinline__ void window_mean(const Sorted& sorted, const WindowArgs& a, int lane, float* __restrict__ out) {
    const int n = a.n, keep = a.keep;
    f32x4 med;
    if (n & 1) {
        med = sorted.at((n - 1) >> 1);
    } else {
        const f32x4 lo = sorted.at((n >> 1) - 1), hi = sorted.at(n >> 1);
#pragma unroll
        for (int c = 0; c < 4; ++c) med[c] = __fmul_rn(__fadd_rn(lo[c], hi[c]), 0.5f);
    }
    if (keep <= 0) {  // np.mean([]) -> nan
        if (lane < 4 && a.col0 + lane < a.n_cols) out[a.col0 + lane] = __uint_as_float(0x7fc00000u);
        return;
    }
    // lo[c] = how many of the lowest ranks fall outside the k nearest neighbours of the median
    const int n_drop = n - keep;
    int lo[4] = {0, 0, 0, 0};
    for (int base = 0; base < n_drop; base += 64) {
        const int i = base + lane;
        const bool active = i < n_drop;
        const f32x4 left = sorted.at(active ? i : 0), right = sorted.at(active ? i + keep : 0);
        const int dl = abs(2 * i - (n - 1)), dr = abs(2 * (i + keep) - (n - 1));
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float al = fabsf(__fsub_rn(left[c], med[c])), ar = fabsf(__fsub_rn(right[c], med[c]));
            const bool farther = active && (al > ar || (al == ar && dl > dr));
            lo[c] += __popcll(__ballot(farther));
        }
    }
    // sum of the kept deviations
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const f32x4 v = sorted.at(i < n ? i : 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool in = i < n && static_cast<unsigned>(i - lo[c]) < static_cast<unsigned>(keep);
            sum[c] += in ? __fsub_rn(v[c], med[c]) : 0.0f;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) sum[c] += __shfl_xor(sum[c], m, 64);
    }
    // exact cross-side ties at the window edge: the reference's stable sort decides by row order
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int first = lo[c], last = lo[c] + keep - 1;
        const float m = med[c];
        const float d_first = __fsub_rn(quad_get(sorted.at(first), c), m);
        const float d_last = __fsub_rn(quad_get(sorted.at(last), c), m);
        bool tie = false;
        if (first > 0) {
            const float d_out = __fsub_rn(quad_get(sorted.at(first - 1), c), m);
            tie = tie || (fabsf(d_out) == fabsf(d_last) && d_out < 0.0f && d_last > 0.0f);
        }
        if (last + 1 < n) {
            const float d_out = __fsub_rn(quad_get(sorted.at(last + 1), c), m);
            tie = tie || (fabsf(d_out) == fabsf(d_first) && d_first < 0.0f && d_out > 0.0f);
        }
        if (tie && a.col0 + c < a.n_cols) {  // wave-uniform
            const float edge = fmaxf(fabsf(d_first), fabsf(d_last));
            float closer = 0.0f;
            int n_closer = 0;
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                const float d = i < n ? __fsub_rn(quad_get(sorted.at(i), c), m) : 0.0f;
                const bool in = i < n && fabsf(d) < edge;
                closer += in ? d : 0.0f;
                n_closer += __popcll(__ballot(in));
            }
#pragma unroll
            for (int mm = 32; mm > 0; mm >>= 1) closer += __shfl_xor(closer, mm, 64);
            int want = keep - n_closer, taken = 0, n_pos = 0, n_neg = 0;
            for (int base = 0; base < n && taken < want; base += 64) {
                const int r = base + lane;
                float d = 0.0f;
                bool tied = false;
                if (r < n) {
                    const int64_t src = a.row_index ? a.row_index[r] : r;
                    d = __fsub_rn(a.G[src * a.ld + a.col0 + c], m);
                    tied = fabsf(d) == edge;
                }
                const unsigned long long mask = __ballot(tied);
                const int before = __popcll(mask & ((1ull << lane) - 1ull));
                const bool take = tied && (taken + before < want);
                n_pos += __popcll(__ballot(take && d > 0.0f));
                n_neg += __popcll(__ballot(take && d < 0.0f));
                taken += __popcll(mask);
            }
            sum[c] = closer + static_cast<float>(n_pos - n_neg) * edge;
        }
    }
    if (lane < 4 && a.col0 + lane < a.n_cols) {
        const float s = lane == 0 ? sum[0] : (lane == 1 ? sum[1] : (lane == 2 ? sum[2] : sum[3]));
        const float m = lane == 0 ? med.x : (lane == 1 ? med.y : (lane == 2 ? med.z : med.w));
        out[a.col0 + lane] = __fadd_rn(__fdiv_rn(s, static_cast<float>(keep)), m);
    }
}

// ---- general kernel: LDS bitonic over [rank][V columns] ------------------------------------------
// V = 4 columns per pass up to 8192 rows (128 KiB of LDS); V = 2 up to 16,384 rows (round 3: the reference has no row limit,
// defences.py:44-52, and configs[4]'s N = 10,000 client matrix used to come back BYZ_E_UNSUPPORTED).  A column of more
// than 5376 rows is not a case any BASELINE configuration times: this path is about being there, not about speed.
template <int V>
struct ColumnArray {
    typedef float vec_t __attribute__((ext_vector_type(V)));
    const vec_t* base;
    __device__ __forceinline__ f32x4 at(int rank) const {
        const vec_t v = base[rank];
        if constexpr (V == 4) return f32x4{v[0], v[1], v[2], v[3]};
        else return f32x4{v[0], v[1], v[0], v[1]};   // (columns 2, 3 of the quad are never written: see n_cols below)
    }
};

template <int V>
__global__ __launch_bounds__(1024) void trimmed_mean_lds_kernel(const float* __restrict__ G, int n_rows, int n_pad,
                                                                int64_t n_cols, int64_t ld,
                                                                const int32_t* __restrict__ row_index, int keep,
                                                                float* __restrict__ out) {
    typedef float vec_t __attribute__((ext_vector_type(V)));
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];  // n_pad entries of V floats
    vec_t* const quad = reinterpret_cast<vec_t*>(lds_raw);
    __shared__ int nan_columns;   // bit e: column c + e holds a NaN (np.median makes the whole result NaN then)
    const int tid = threadIdx.x, nt = blockDim.x;
    const int64_t n_quads = (n_cols + V - 1) / V;
    const float pinf = __builtin_inff();
    for (int64_t qd = blockIdx.x; qd < n_quads; qd += gridDim.x) {
        const int64_t c = qd * V;
        if (tid == 0) nan_columns = 0;
        __syncthreads();
        int seen_nan = 0;
        for (int r = tid; r < n_pad; r += nt) {
            vec_t v;
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = pinf;
            if (r < n_rows) {
                const int64_t src = row_index ? row_index[r] : r;
                const float* p = G + src * ld + c;
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] = c + e < n_cols ? p[e] : 0.0f;
                // this file is compiled with -fno-honor-nans (the sorting network's min / max): test the bits, not x != x
#pragma unroll
                for (int e = 0; e < V; ++e)
                    seen_nan |= (__float_as_uint(v[e]) & 0x7fffffffu) > 0x7f800000u ? (1 << e) : 0;
            }
            quad[r] = v;
        }
        if (seen_nan != 0) atomicOr(&nan_columns, seen_nan);
        __syncthreads();
        for (int k = 2; k <= n_pad; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int idx = tid; idx < (n_pad >> 1); idx += nt) {
                    const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                    const int p = i | j;
                    const vec_t a = quad[i], b = quad[p];
                    const bool up = (i & k) == 0;
                    vec_t lo, hi;
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        lo[e] = __builtin_fminf(a[e], b[e]);
                        hi[e] = __builtin_fmaxf(a[e], b[e]);
                    }
                    quad[i] = up ? lo : hi;
                    quad[p] = up ? hi : lo;
                }
                __syncthreads();
            }
        }
        if (tid < 64) {
            const ColumnArray<V> sorted{quad};
            // (V = 2: the window code works on quads; columns beyond c + V are cut off through its column bound)
            const WindowArgs args{G, ld, row_index, n_rows, keep, c, n_cols < c + V ? n_cols : c + V};
            window_mean(sorted, args, tid, out);
            // a NaN anywhere in the column: the reference's np.median is NaN and so is everything after it; the sorted
            // order above is unspecified for such a column, the result is not
            if (tid < V && ((nan_columns >> tid) & 1) && c + tid < n_cols) out[c + tid] = __uint_as_float(0x7fc00000u);
        }
        __syncthreads();
    }
}

// ---- more than 16,384 rows: the columns sorted in global memory (large_rows.hip's segment sort), the same window code ----
// Keys: the order-preserving float bits alone (32 bits: the window code asks a sorted column for values only, and settles ties at its
// edge by reading G in row order); padding rows are all-ones keys, behind every value (and decoding to a NaN like one).
__device__ __forceinline__ uint32_t ordered_bits(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {     // (branch-free: a select here crashes this compiler's ISel)
    return __uint_as_float(o ^ (~static_cast<uint32_t>(static_cast<int32_t>(o) >> 31) | 0x80000000u));
}

// 64 rows x 64 columns of the batch per workgroup: read along the rows of G, written along the columns' key arrays (a thread per
// (row, column) read 4 bytes out of every 64-byte sector it touched).  A column beyond the matrix is filled with zeros (its quad's
// other columns are real); a padding row is an all-ones key.
__global__ __launch_bounds__(256) void large_column_keys_kernel(const float* __restrict__ G, int n_rows, int64_t n_pad, int64_t c0,
                                                                int64_t n_batch_cols, int64_t n_cols, int64_t ld,
                                                                const int32_t* __restrict__ row_index, uint32_t* __restrict__ keys) {
    __shared__ uint32_t tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * 64, b0 = static_cast<int64_t>(blockIdx.y) * 64;
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i, c = c0 + b0 + tx;
        uint32_t key = ~0u;
        if (r < n_rows) {
            const int64_t src = row_index ? row_index[r] : r;
            key = ordered_bits(c < n_cols ? G[src * ld + c] : 0.0f);
        }
        tile[i][tx] = key;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int64_t b = b0 + i, r = r0 + tx;
        if (b < n_batch_cols && r < n_pad) keys[b * n_pad + r] = tile[tx][i];
    }
}

struct SortedKeyQuad {
    const uint32_t* base;   // four columns of n_pad sorted keys each
    int64_t n_pad;
    __device__ __forceinline__ f32x4 at(int rank) const {
        return f32x4{from_ordered_bits(base[rank]), from_ordered_bits(base[n_pad + rank]), from_ordered_bits(base[2 * n_pad + rank]),
                     from_ordered_bits(base[3 * n_pad + rank])};
    }
};

// one wave per quad of columns
__global__ __launch_bounds__(64) void large_window_kernel(const uint32_t* __restrict__ keys, int64_t n_pad,
                                                          const float* __restrict__ G, int64_t ld,
                                                          const int32_t* __restrict__ row_index, int n_rows, int keep, int64_t c0,
                                                          int64_t n_cols, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int64_t c = c0 + 4 * static_cast<int64_t>(blockIdx.x);
    const SortedKeyQuad sorted{keys + 4 * static_cast<int64_t>(blockIdx.x) * n_pad, n_pad};
    const WindowArgs args{G, ld, row_index, n_rows, keep, c, n_cols < c + 4 ? n_cols : c + 4};
    window_mean(sorted, args, lane, out);
    // a NaN anywhere in the column: np.median is NaN and so is everything after it.  By the keys a NaN sorts to one of the
    // two ends (sign bit set: first, clear: last)
    if (lane < 4 && c + lane < n_cols) {
        const uint32_t first = __float_as_uint(from_ordered_bits(sorted.base[lane * n_pad]));
        const uint32_t last = __float_as_uint(from_ordered_bits(sorted.base[lane * n_pad + n_rows - 1]));
        if ((first & 0x7fffffffu) > 0x7f800000u || (last & 0x7fffffffu) > 0x7f800000u) out[c + lane] = __uint_as_float(0x7fc00000u);
    }
}

__global__ void lane_selftest_kernel(int32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int masks[11] = {1, 2, 3, 4, 7, 8, 15, 16, 31, 32, 63};
    const float me = static_cast<float>(lane);
#pragma unroll
    for (int p = 0; p < 11; ++p) out[p * 64 + lane] = static_cast<int32_t>(lane_xor(me, masks[p], lane));
}

}  // namespace

int64_t trimmed_mean_max_rows() { return kLargeMaxRows; }

bool trimmed_mean_large_applies(int64_t n_rows) {
    if (n_rows > kGeneralMaxRows) return true;
    const char* e = std::getenv("BYZ_TM_LARGE");     // (read per call: the tests flip it inside one process)
    return e != nullptr && std::atoi(e) != 0;
}

// defences.py:44-52 for any number of rows: batches of columns, each column's values sorted as 64-bit keys in global memory,
// the window found from the sorted column exactly as the LDS kernel below finds it
int launch_trimmed_mean_large(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                              int64_t keep, float* out, hipStream_t stream) {
    if (n_rows > kLargeMaxRows) {
        set_error("trimmed_mean supports at most %lld rows, got %lld", (long long)kLargeMaxRows, (long long)n_rows);
        return BYZ_E_UNSUPPORTED;
    }
    const int64_t n_pad = next_pow2(n_rows < 2 ? 2 : n_rows);
    int64_t batch = static_cast<int64_t>(large_key_scratch_bytes() / (static_cast<size_t>(n_pad) * 4)) & ~int64_t{3};
    if (batch < 4) batch = 4;
    if (batch > 32768) batch = 32768;                       // (the keys kernel's grid.y)
    const int64_t cols4 = ceil_div(n_cols, 4) * 4;
    if (batch > cols4) batch = cols4;
    BYZ_TRY(ctx->large_keys.ensure(static_cast<size_t>(batch) * n_pad * 4));
    uint32_t* keys = ctx->large_keys.as<uint32_t>();
    for (int64_t c0 = 0; c0 < n_cols; c0 += batch) {
        const int64_t cols = cols4 - c0 < batch ? cols4 - c0 : batch;     // a multiple of 4
        large_column_keys_kernel<<<dim3(static_cast<unsigned>(ceil_div(n_pad, 64)), static_cast<unsigned>(ceil_div(cols, 64))), 256, 0, stream>>>(
            G, (int)n_rows, n_pad, c0, cols, n_cols, ld, row_index, keys);
        BYZ_TRY(check_launch("large_column_keys_kernel"));
        BYZ_TRY(segment_sort_u32(ctx, keys, cols, n_pad, stream));
        large_window_kernel<<<static_cast<unsigned>(cols / 4), 64, 0, stream>>>(keys, n_pad, G, ld, row_index, (int)n_rows, (int)keep, c0,
                                                                             n_cols, out);
        BYZ_TRY(check_launch("large_window_kernel"));
    }
    return BYZ_OK;
}

int launch_lane_selftest(byz_ctx* ctx, int32_t* out, int32_t* n_patterns, hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    lane_selftest_kernel<<<1, 64, 0, stream>>>(out);
    *n_patterns = 11;
    return check_launch("lane_selftest_kernel");
}

int launch_trimmed_mean_sorted(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                               const int32_t* row_index, int64_t keep, float* out, hipStream_t stream) {
    if (n_rows > kGeneralMaxRows) return launch_trimmed_mean_large(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
    const int64_t n_pad = next_pow2(n_rows);
    const int vec = n_pad <= 8192 ? 4 : 2;
    const int64_t n_quads = ceil_div(n_cols, vec);
    const int64_t grid = n_quads < static_cast<int64_t>(ctx->num_cus) * 2 ? n_quads : static_cast<int64_t>(ctx->num_cus) * 2;
    const size_t lds = static_cast<size_t>(n_pad) * vec * sizeof(float);
    if (vec == 4) {
        BYZ_HIP(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&trimmed_mean_lds_kernel<4>), static_cast<int>(lds)));
        trimmed_mean_lds_kernel<4><<<static_cast<unsigned>(grid), 1024, lds, stream>>>(G, (int)n_rows, (int)n_pad, n_cols, ld,
                                                                                       row_index, (int)keep, out);
    } else {
        BYZ_HIP(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&trimmed_mean_lds_kernel<2>), static_cast<int>(lds)));
        trimmed_mean_lds_kernel<2><<<static_cast<unsigned>(grid), 1024, lds, stream>>>(G, (int)n_rows, (int)n_pad, n_cols, ld,
                                                                                       row_index, (int)keep, out);
    }
    return check_launch("trimmed_mean_lds_kernel");
}

}  // namespace byz
