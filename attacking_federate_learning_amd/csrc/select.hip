// Krum / Bulyan selection on the N x N distance matrix.
//
//   krum          reference defences.py:23-42   one scoring pass + argmin
//   bulyan        reference defences.py:55-68   theta = n - 2f dependent picks with removal
//
// row_sort_kernel (one workgroup per row): ascending bitonic sort of the row's N-1 distances in LDS as
// 64-bit keys (order-preserving float bits << 32 | column), then
//   * the Krum score exactly as the reference forms it: a *sequential* fp32 sum of the first
//     `prefix_len` sorted values (Python sum over np.float32 scalars, defences.py:33-34).  Given the same
//     distance matrix the scores, and therefore the winner, are bit-identical to the reference's;
//   * for Bulyan: the sorted column order, the transposed rank table and two fp64 sums per row.
//
// krum_argmin_kernel: the loop of defences.py:27-37 as a reduction -- candidates compared in the
// reference's visit order 1, 0, 2, 3, ... with a strict '<' (a tie keeps the earlier visitor), against a
// running minimum that starts at 1e20 (no score below it -> index -1).
//
// bulyan_loop_kernel: one persistent 1024-thread workgroup runs all theta picks on the device, no host
// round trips.  Re-sorting every remaining row per pick, as the reference does, costs O(theta N^2 log N);
// here each row keeps two running fp64 sums: T = sum of its distances to the rows still present, and
// Top = sum of the `drop` largest of them (drop = f - 1 when users_count == N).  The reference's score,
// "sum of the n_t - f smallest of the n_t - 1 remaining distances", is T - Top, and removing the winner w
// updates both in O(1) per row through the precomputed rank table.  All the terms are fp32 values, so
// the fp64 sums are exact for any realistic spread of magnitudes: rows with identical distance multisets
// (the identical malicious vectors) keep bitwise identical scores and tie exactly as in the reference,
// where the visit order decides.  Versus the reference's fp32 sequential sums the scores differ by fp32
// rounding noise (~1e-6 relative at N = 1e4); DESIGN.md states the parity protocol for that.
#include "common.hpp"

namespace byz {
namespace {

constexpr int kMaxSelectRows = 16384;  // 128 KiB of 64-bit keys in LDS
constexpr float kKrumInit = 1e20f;     // defences.py:27

__device__ __forceinline__ uint32_t ordered_bits(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ int visit_position(int u) { return u == 0 ? 1 : (u == 1 ? 0 : u); }

// ---------------------------------------------------------------------------------------------------
template <bool TABLES>
__global__ void row_sort_kernel(const float* __restrict__ dist, int n, int n_pad, int prefix_len, int drop,
                                float* __restrict__ scores, uint16_t* __restrict__ sorted_idx,
                                uint16_t* __restrict__ rank_t, double* __restrict__ row_total,
                                double* __restrict__ row_top) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // n_pad keys, then scratch
    double* scratch = reinterpret_cast<double*>(keys + n_pad);                 // blockDim.x doubles
    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    const int nt = blockDim.x;

    const float* row = dist + static_cast<int64_t>(u) * n;
    for (int c = tid; c < n_pad; c += nt) {
        unsigned long long key;
        if (c >= n) {
            key = ~0ull;
        } else {
            const float d = (c == u) ? __builtin_inff() : row[c];  // self entry sorts behind every real one
            key = (static_cast<unsigned long long>(ordered_bits(d)) << 32) | static_cast<unsigned>(c);
        }
        keys[c] = key;
    }
    __syncthreads();

    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < (n_pad >> 1); idx += nt) {
                const int i = ((idx & ~(j - 1)) << 1) | (idx & (j - 1));
                const int p = i | j;
                const unsigned long long a = keys[i], b = keys[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            __syncthreads();
        }
    }
    // A self distance of +inf can tie with a genuine +inf/NaN entry only for poisoned input; otherwise the
    // real neighbours occupy ranks 0 .. n-2.

    if (tid == 0) {
        float s = 0.0f;
        for (int r = 0; r < prefix_len; ++r) s = __fadd_rn(s, from_ordered_bits(static_cast<uint32_t>(keys[r] >> 32)));
        scores[u] = s;
    }

    if (TABLES) {
        double tot = 0.0, top = 0.0;
        const int first_top = n - 1 - drop;
        for (int r = tid; r < n; r += nt) {
            const unsigned long long key = keys[r];
            const int c = static_cast<int>(key & 0xffffffffu);
            sorted_idx[static_cast<int64_t>(u) * n + r] = static_cast<uint16_t>(c);
            rank_t[static_cast<int64_t>(c) * n + u] = static_cast<uint16_t>(r);
            if (r < n - 1) {
                const double v = static_cast<double>(from_ordered_bits(static_cast<uint32_t>(key >> 32)));
                tot += v;
                if (r >= first_top) top += v;
            }
        }
        // fixed-shape tree: identical sorted rows reduce to identical sums
        scratch[tid] = tot;
        __syncthreads();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) scratch[tid] += scratch[tid + s];
            __syncthreads();
        }
        if (tid == 0) row_total[u] = scratch[0];
        __syncthreads();
        scratch[tid] = top;
        __syncthreads();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) scratch[tid] += scratch[tid + s];
            __syncthreads();
        }
        if (tid == 0) row_top[u] = scratch[0];
    }
}

// ---------------------------------------------------------------------------------------------------
struct Candidate {
    double score;
    int pos;   // position in the reference's visit order; INT_MAX = none
    int row;
};

__device__ __forceinline__ bool better(const Candidate& a, const Candidate& b) {
    // strict '<' on the score; an equal score keeps the earlier visitor
    return a.score < b.score || (a.score == b.score && a.pos < b.pos);
}

__device__ __forceinline__ Candidate wave_best(Candidate c) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        Candidate o;
        o.score = __shfl_xor(c.score, m, 64);
        o.pos = __shfl_xor(c.pos, m, 64);
        o.row = __shfl_xor(c.row, m, 64);
        if (better(o, c)) c = o;
    }
    return c;
}

// Block-wide best candidate, broadcast to every thread.  `slots` holds blockDim.x/64 candidates.
__device__ __forceinline__ Candidate block_best(Candidate c, Candidate* slots) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    c = wave_best(c);
    __syncthreads();  // slots may still be read from the previous call
    if (lane == 0) slots[wave] = c;
    __syncthreads();
    Candidate best = slots[0];
    for (int w = 1; w < n_waves; ++w) {
        const Candidate o = slots[w];
        if (better(o, best)) best = o;
    }
    return best;
}

__global__ __launch_bounds__(1024) void krum_argmin_kernel(const float* __restrict__ scores, int n,
                                                           int32_t* __restrict__ winner) {
    __shared__ Candidate slots[16];
    Candidate c{static_cast<double>(kKrumInit), 0x7fffffff, -1};
    for (int u = threadIdx.x; u < n; u += blockDim.x) {
        const float s = scores[u];
        if (s < kKrumInit) {  // false for NaN, as in the reference's comparison
            Candidate o{static_cast<double>(s), visit_position(u), u};
            if (better(o, c)) c = o;
        }
    }
    // a single row has an empty distance dict in the reference: nothing is visited, index stays -1
    const Candidate best = block_best(c, slots);
    if (threadIdx.x == 0) *winner = (n < 2) ? -1 : best.row;
}

// ---------------------------------------------------------------------------------------------------
template <int Q>  // rows per thread: thread t owns rows t, t + 1024, ...
__global__ __launch_bounds__(1024) void bulyan_loop_kernel(const float* __restrict__ dist, int n, int theta,
                                                           int drop, const uint16_t* __restrict__ sorted_idx,
                                                           const uint16_t* __restrict__ rank_t,
                                                           const double* __restrict__ row_total,
                                                           const double* __restrict__ row_top,
                                                           int32_t* __restrict__ selection,
                                                           int32_t* __restrict__ status) {
    __shared__ Candidate slots[16];
    __shared__ uint32_t removed[kMaxSelectRows / 32];
    const int tid = threadIdx.x;
    for (int i = tid; i < kMaxSelectRows / 32; i += 1024) removed[i] = 0u;

    double tot[Q], top[Q];
    int ptr[Q];
    bool alive[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int u = tid + q * 1024;
        alive[q] = u < n;
        tot[q] = alive[q] ? row_total[u] : 0.0;
        top[q] = (alive[q] && drop > 0) ? row_top[u] : 0.0;
        ptr[q] = n - 1 - drop;
    }
    __syncthreads();

    int failed = 0;
    for (int t = 0; t < theta; ++t) {
        Candidate c{static_cast<double>(kKrumInit), 0x7fffffff, -1};
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (alive[q]) {
                const int u = tid + q * 1024;
                const double s = tot[q] - top[q];
                if (s < static_cast<double>(kKrumInit)) {
                    Candidate o{s, visit_position(u), u};
                    if (better(o, c)) c = o;
                }
            }
        }
        const Candidate best = block_best(c, slots);
        // a single-row matrix has an empty distance dict in the reference (nothing to visit); a last
        // survivor of a larger matrix is still visited with an empty list, scores 0 and is picked
        const int w = (n < 2) ? -1 : best.row;
        if (w < 0) {
            failed = 1;
            break;  // uniform: every thread sees the same broadcast
        }
        if (tid == 0) {
            selection[t] = w;
            removed[w >> 5] |= 1u << (w & 31);
        }
        __syncthreads();
        const float* drow = dist + static_cast<int64_t>(w) * n;        // symmetric: d[w][u] == d[u][w]
        const uint16_t* rrow = rank_t + static_cast<int64_t>(w) * n;   // rank of column w inside row u
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int u = tid + q * 1024;
            if (!alive[q]) continue;
            if (u == w) {
                alive[q] = false;
                continue;
            }
            const double d = static_cast<double>(drow[u]);
            const int r = rrow[u];
            tot[q] -= d;
            if (drop > 0 && r >= ptr[q]) {
                // w was one of this row's `drop` largest: the largest survivor below the boundary joins them
                top[q] -= d;
                int p = ptr[q] - 1;
                const uint16_t* order = sorted_idx + static_cast<int64_t>(u) * n;
                while (p >= 0) {
                    const int col = order[p];
                    if (!((removed[col >> 5] >> (col & 31)) & 1u)) break;
                    --p;
                }
                if (p >= 0) top[q] += static_cast<double>(dist[static_cast<int64_t>(u) * n + order[p]]);
                ptr[q] = p;
            }
        }
    }
    if (tid == 0) *status = failed;
}

}  // namespace

int64_t select_max_rows() { return kMaxSelectRows; }

int launch_row_sort(byz_ctx* ctx, const float* dist, int64_t n, int64_t prefix_len, int64_t drop_count,
                    bool want_tables, hipStream_t stream) {
    BYZ_REQUIRE(dist && n > 0, "row sort: bad arguments");
    if (n > kMaxSelectRows) {
        set_error("selection kernels support at most %d rows, got %lld", kMaxSelectRows, (long long)n);
        return BYZ_E_UNSUPPORTED;
    }
    int64_t n_pad = next_pow2(n);
    if (n_pad < 128) n_pad = 128;
    int threads = static_cast<int>(n_pad / 2);
    if (threads > 1024) threads = 1024;
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(n) * sizeof(float)));
    if (want_tables) {
        BYZ_TRY(ctx->sorted_idx.ensure(static_cast<size_t>(n) * n * sizeof(uint16_t)));
        BYZ_TRY(ctx->rank_t.ensure(static_cast<size_t>(n) * n * sizeof(uint16_t)));
        BYZ_TRY(ctx->row_total.ensure(static_cast<size_t>(n) * sizeof(double)));
        BYZ_TRY(ctx->row_top.ensure(static_cast<size_t>(n) * sizeof(double)));
    }
    const size_t lds = static_cast<size_t>(n_pad) * 8 + static_cast<size_t>(threads) * 8;
    KernelTimer t(ctx, BYZ_K_ROW_SORT, stream);
    if (want_tables) {
        BYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&row_sort_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        row_sort_kernel<true><<<static_cast<unsigned>(n), threads, lds, stream>>>(
            dist, (int)n, (int)n_pad, (int)prefix_len, (int)drop_count, ctx->scores.as<float>(),
            ctx->sorted_idx.as<uint16_t>(), ctx->rank_t.as<uint16_t>(), ctx->row_total.as<double>(),
            ctx->row_top.as<double>());
    } else {
        BYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&row_sort_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        row_sort_kernel<false><<<static_cast<unsigned>(n), threads, lds, stream>>>(
            dist, (int)n, (int)n_pad, (int)prefix_len, (int)drop_count, ctx->scores.as<float>(), nullptr,
            nullptr, nullptr, nullptr);
    }
    return check_launch("row_sort_kernel");
}

int launch_krum_argmin(byz_ctx* ctx, int64_t n, int32_t* winner_dev, hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_KRUM_ARGMIN, stream);
    krum_argmin_kernel<<<1, 1024, 0, stream>>>(ctx->scores.as<float>(), (int)n, winner_dev);
    return check_launch("krum_argmin_kernel");
}

int launch_bulyan_loop(byz_ctx* ctx, const float* dist, int64_t n, int64_t theta, int64_t drop_count,
                       int32_t* selection_dev, int32_t* status_dev, hipStream_t stream) {
    BYZ_REQUIRE(dist && selection_dev && status_dev && n > 0 && theta >= 0 && theta <= n,
                "bulyan loop: bad arguments (n=%lld theta=%lld)", (long long)n, (long long)theta);
    KernelTimer t(ctx, BYZ_K_BULYAN_LOOP, stream);
    const uint16_t* si = ctx->sorted_idx.as<uint16_t>();
    const uint16_t* rt = ctx->rank_t.as<uint16_t>();
    const double* tot = ctx->row_total.as<double>();
    const double* top = ctx->row_top.as<double>();
    const int q = static_cast<int>(ceil_div(n, 1024));
#define BYZ_LOOP(Q) bulyan_loop_kernel<Q><<<1, 1024, 0, stream>>>(dist, (int)n, (int)theta, (int)drop_count, si, rt, tot, top, selection_dev, status_dev)
    if (q <= 1) BYZ_LOOP(1);
    else if (q <= 2) BYZ_LOOP(2);
    else if (q <= 4) BYZ_LOOP(4);
    else if (q <= 8) BYZ_LOOP(8);
    else BYZ_LOOP(16);
#undef BYZ_LOOP
    return check_launch("bulyan_loop_kernel");
}

}  // namespace byz
