// More rows than the LDS-resident kernels hold (select.hip, trimmed_mean.hip: 16,384).  The reference has no limit
// (defences.py:23-70 are loops over Python lists); no BASELINE configuration goes beyond N = 10,000, so this file is about
// BEING THERE with the reference's results, not about speed: textbook kernels on global memory, 32-bit indices throughout.
//
//   segment_sort_u64            many independent arrays of 64-bit keys, each a power of two long, sorted ascending: a bitonic
//                               network whose merges of up to 4096 keys run in LDS and whose longer strides are one launch each
//   launch_row_sort_large       defences.py:31-34: every row of the N x N distance matrix sorted (the same keys as select.hip:
//                               order-preserving float bits << 32 | column, the self entry last, a NaN behind +inf), the Krum
//                               score as the SEQUENTIAL fp32 sum of the first `prefix_len` sorted values; for Bulyan the sorted
//                               values, the sorted columns, the rank of every column and two fp64 sums per row
//   launch_bulyan_loop_large    defences.py:59-68, two launches per pick.  A row's exact score -- the sum of all its live
//                               distances minus the sum of the `drop` largest, both carried in fp64 and updated in O(1) when a
//                               row leaves -- bounds the reference's sequential fp32 sum from both sides ((1 +- u)^(m-1),
//                               u = 2^-24); every live row whose lower bound does not exceed the smallest upper bound is a
//                               CONTENDER and is scored again exactly the reference's way (a walk along its sorted row that
//                               skips the rows already picked and adds the first m live values left to right in fp32); the
//                               reference's strict '<' in its visit order 1, 0, 2, ... then decides among the contenders.  A row
//                               outside the band cannot win or tie, so the selection is the reference's, pick for pick.  Rows
//                               with a negative or non-finite entry (an arbitrary caller-supplied matrix) always contend.
#include "common.hpp"

#include <cstdlib>

namespace byz {
namespace {

constexpr int kChunk = 4096;          // keys of one LDS-resident merge (32 KiB)
constexpr int kSortThreads = 1024;
constexpr float kKrumInit = 1e20f;    // defences.py:27
constexpr size_t kKeyScratchBytes = size_t{4} << 30;

__device__ __forceinline__ uint32_t ordered_bits(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ int visit_position(int u) { return u == 0 ? 1 : (u == 1 ? 0 : u); }

// FIRST: every merge k = 2 .. len of one chunk; else the levels j = len / 2 .. 1 of the merge of size k_merge
template <bool FIRST>
__global__ __launch_bounds__(kSortThreads) void segment_sort_local_kernel(unsigned long long* __restrict__ keys, int64_t n_pad,
                                                                          int chunks_per_seg, int64_t k_merge) {
    __shared__ unsigned long long lds[kChunk];
    const int64_t seg = blockIdx.x / chunks_per_seg;
    const int64_t ch = blockIdx.x % chunks_per_seg;
    const int len = n_pad < kChunk ? static_cast<int>(n_pad) : kChunk;
    const int64_t i0 = ch * kChunk;                       // position of lds[0] inside its segment
    unsigned long long* const base = keys + seg * n_pad + i0;
    const int tid = threadIdx.x;
    for (int i = tid; i < len; i += kSortThreads) lds[i] = base[i];
    __syncthreads();
    for (int64_t k = FIRST ? 2 : k_merge; k <= (FIRST ? static_cast<int64_t>(len) : k_merge); k <<= 1) {
        const int j_top = k / 2 < len / 2 ? static_cast<int>(k / 2) : len / 2;
        for (int j = j_top; j > 0; j >>= 1) {
            for (int p = tid; p < len / 2; p += kSortThreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int q = i | j;
                const unsigned long long a = lds[i], b = lds[q];
                const bool up = ((i0 + i) & k) == 0;
                if ((a > b) == up) {
                    lds[i] = b;
                    lds[q] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < len; i += kSortThreads) base[i] = lds[i];
}

// one level (stride j >= kChunk) of the merge of size k, every segment at once
__global__ __launch_bounds__(256) void segment_sort_global_kernel(unsigned long long* __restrict__ keys, int64_t n_pad,
                                                                  int64_t n_pairs, int64_t k, int64_t j) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (g >= n_pairs) return;
    const int64_t half = n_pad >> 1;
    const int64_t seg = g / half, p = g % half;
    const int64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
    const int64_t q = i | j;
    unsigned long long* const base = keys + seg * n_pad;
    const unsigned long long a = base[i], b = base[q];
    const bool up = (i & k) == 0;
    if ((a > b) == up) {
        base[i] = b;
        base[q] = a;
    }
}

// ---- the row sort -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void large_row_keys_kernel(const float* __restrict__ dist, int n, int64_t n_pad, int row0,
                                                             unsigned long long* __restrict__ keys) {
    const int u = row0 + blockIdx.y;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (c >= n_pad) return;
    unsigned long long key = ~0ull;
    if (c < n) {
        // the self entry sorts behind every real one, +inf and NaN included; a NaN of either sign behind +inf (select.hip)
        const float d = dist[static_cast<int64_t>(u) * n + c];
        const uint32_t ob = (c == u) ? 0xffffffffu : (d != d ? 0xfffffffeu : ordered_bits(d));
        key = (static_cast<unsigned long long>(ob) << 32) | static_cast<unsigned>(c);
    }
    keys[static_cast<int64_t>(blockIdx.y) * n_pad + c] = key;
}

// the per-row state of the Bulyan loop (ctx->large_state): doubles first, then the 32-bit words, then the bytes
struct LargeState {
    double* total;        // [n] sum of the row's live distances (regular rows)
    double* top;          // [n] sum of its `drop` largest live distances
    double* scale;        // [1] the largest first total: what the absolute rounding slack of the running sums is taken from
    int32_t* top_first;   // [n] the lowest rank that belongs to the `drop` largest live entries
    int32_t* irregular;   // [n] the row holds a negative or non-finite distance: it is scored the reference's way at every pick
    int32_t* contender;   // [n] the rows to be scored the reference's way at the current pick
    int32_t* n_contenders;// [1]
    float* score;         // [n] their scores
    uint8_t* gone;        // [n] picked already
};
__host__ __device__ inline size_t large_state_bytes(int64_t n) {
    return static_cast<size_t>(2 * n + 2) * sizeof(double) + static_cast<size_t>(4 * n + 2) * sizeof(int32_t) + static_cast<size_t>(n) + 64;
}
__host__ __device__ inline LargeState large_state(void* p, int64_t n) {
    LargeState s;
    s.total = static_cast<double*>(p);
    s.top = s.total + n;
    s.scale = s.top + n;
    s.top_first = reinterpret_cast<int32_t*>(s.scale + 2);
    s.irregular = s.top_first + n;
    s.contender = s.irregular + n;
    s.n_contenders = s.contender + n;
    s.score = reinterpret_cast<float*>(s.n_contenders + 2);
    s.gone = reinterpret_cast<uint8_t*>(s.score + n);
    return s;
}

// One workgroup per sorted row: the Krum score, and for Bulyan the tables and the row's state.
template <bool TABLES>
__global__ __launch_bounds__(256) void large_row_tables_kernel(const unsigned long long* __restrict__ keys, int n, int64_t n_pad,
                                                               int row0, int prefix_len, int drop, float* __restrict__ scores,
                                                               float* __restrict__ sorted_val, uint32_t* __restrict__ sorted_idx,
                                                               uint32_t* __restrict__ rank_rows, LargeState st) {
    __shared__ double red[256];
    const int u = row0 + blockIdx.x;
    const int tid = threadIdx.x;
    const unsigned long long* const row = keys + static_cast<int64_t>(blockIdx.x) * n_pad;
    if (tid < 64) {
        // Python's sum() over np.float32 scalars (defences.py:34): 0 + x0, then one fp32 addition per value, left to right
        float s = 0.0f;
        for (int r0 = 0; r0 < prefix_len; r0 += 64) {
            const int r = r0 + tid;
            const float v = r < prefix_len ? from_ordered_bits(static_cast<uint32_t>(row[r] >> 32)) : 0.0f;
            const int cnt = prefix_len - r0 < 64 ? prefix_len - r0 : 64;
            for (int i = 0; i < cnt; ++i) s = __fadd_rn(s, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)));
        }
        if (tid == 0) scores[u] = s;
    }
    if constexpr (TABLES) {
        double tot = 0.0, top = 0.0;
        int odd = 0;
        const int first_top = n - 1 - drop;
        for (int r = tid; r < n; r += 256) {
            const unsigned long long key = row[r];
            const int c = static_cast<int>(key & 0xffffffffu);
            const float v = c == u ? __builtin_inff() : from_ordered_bits(static_cast<uint32_t>(key >> 32));
            sorted_val[static_cast<int64_t>(u) * n + r] = v;
            sorted_idx[static_cast<int64_t>(u) * n + r] = static_cast<uint32_t>(c);
            rank_rows[static_cast<int64_t>(u) * n + c] = static_cast<uint32_t>(r);
            if (c != u) {
                if (v >= 0.0f && v <= 3.4028234663852886e38f) {
                    tot += static_cast<double>(v);
                    if (r >= first_top) top += static_cast<double>(v);
                } else {
                    odd = 1;
                }
            }
        }
        // fixed-shape trees
        red[tid] = tot;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) st.total[u] = red[0];
        __syncthreads();
        red[tid] = top;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) st.top[u] = red[0];
        __syncthreads();
        red[tid] = static_cast<double>(odd);
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) {
            st.irregular[u] = red[0] != 0.0 ? 1 : 0;
            st.top_first[u] = first_top;       // (the self entry holds rank n - 1: the `drop` entries below it are the largest)
            st.gone[u] = 0;
        }
    }
}

// ---- the Bulyan loop ----------------------------------------------------------------------------------------------------
struct Pick {
    float score;
    int pos;
    int row;
};
__device__ __forceinline__ bool better(const Pick& a, const Pick& b) {
    return a.score < b.score || (a.score == b.score && a.pos < b.pos);   // strict '<'; an equal score keeps the earlier visitor
}

// how many live values pick t adds per row: users_count - len(selection_set) - corrupted_count of the n - t - 1 that are left
// (defences.py:26, 34, 61)
__device__ __forceinline__ int values_per_row(int n, int t, int users_count, int corrupted) {
    const int want = users_count - t - corrupted, left = n - t - 1;
    return want < left ? (want > 0 ? want : 0) : left;
}

// One workgroup.  Closes pick t - 1 (the reference's comparison loop over the contenders' scores, the winner's removal from
// every row's sums) and opens pick t (exact scores, their minimum, the contenders).
__global__ __launch_bounds__(1024) void large_pick_kernel(int t, int n, int theta, int drop, int users_count, int corrupted,
                                                          const float* __restrict__ sorted_val, const uint32_t* __restrict__ sorted_idx,
                                                          const uint32_t* __restrict__ rank_rows, LargeState st,
                                                          int32_t* __restrict__ selection, int32_t* __restrict__ status,
                                                          int32_t* __restrict__ rescored) {
    __shared__ Pick picks[1024];
    __shared__ double mins[1024];
    __shared__ int n_listed;
    const int tid = threadIdx.x;
    if (*status != 0) return;
    if (t > 0) {
        const int listed = *st.n_contenders;
        Pick mine{kKrumInit, 0x7fffffff, -1};
        for (int i = tid; i < listed; i += 1024) {
            const int u = st.contender[i];
            const float s = st.score[u];
            if (s < kKrumInit) {   // false for NaN, as in the reference's comparison
                const Pick o{s, visit_position(u), u};
                if (better(o, mine)) mine = o;
            }
        }
        picks[tid] = mine;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if (tid < s && better(picks[tid + s], picks[tid])) picks[tid] = picks[tid + s];
            __syncthreads();
        }
        const int w = picks[0].row;
        __syncthreads();
        if (w < 0) {   // minimal_error_index stays -1: the reference's distances.pop(-1) raises KeyError
            if (tid == 0) *status = 1;
            return;
        }
        if (tid == 0) {
            selection[t - 1] = w;
            st.gone[w] = 1;
        }
        __threadfence_block();
        __syncthreads();
        for (int u = tid; u < n; u += 1024) {
            if (st.gone[u] || st.irregular[u]) continue;
            const int64_t row = static_cast<int64_t>(u) * n;
            const int r = static_cast<int>(rank_rows[row + w]);
            const double v = static_cast<double>(sorted_val[row + r]);
            st.total[u] -= v;
            if (drop > 0 && r >= st.top_first[u]) {
                // the winner was one of the `drop` largest: the next live entry below them takes its place
                int p = st.top_first[u] - 1;
                while (p >= 0 && st.gone[sorted_idx[row + p]]) --p;
                st.top[u] += (p >= 0 ? static_cast<double>(sorted_val[row + p]) : 0.0) - v;
                st.top_first[u] = p;
            }
        }
        __syncthreads();
    }
    if (t >= theta) return;
    if (t == 0) {
        double big = 0.0;
        for (int u = tid; u < n; u += 1024) big = st.irregular[u] ? big : (st.total[u] > big ? st.total[u] : big);
        mins[tid] = big;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if (tid < s && mins[tid + s] > mins[tid]) mins[tid] = mins[tid + s];
            __syncthreads();
        }
        if (tid == 0) *st.scale = mins[0];
        __syncthreads();
    }
    const int m = values_per_row(n, t, users_count, corrupted);
    const bool all_of_them = users_count - t - corrupted >= n - t - 1;     // nothing is dropped: the score is the whole total
    double low = __builtin_inf();
    for (int u = tid; u < n; u += 1024) {
        if (st.gone[u] || st.irregular[u]) continue;
        const double s = all_of_them ? st.total[u] : st.total[u] - st.top[u];
        low = s < low ? s : low;
    }
    mins[tid] = low;
    if (tid == 0) n_listed = 0;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s && mins[tid + s] < mins[tid]) mins[tid] = mins[tid + s];
        __syncthreads();
    }
    low = mins[0];
    // fl(sum) of m non-negative terms added one by one lies in [s (1 - u)^(m-1), s (1 + u)^(m-1)], u = 2^-24: a row can reach
    // the smallest sum only from below s_min ((1 + u) / (1 - u))^(m-1) <= s_min (1 + 2.1 m u) for m u <= 2^-6; the running fp64
    // sums carry at most 2^17 roundings of 2^-53 of the largest first total each.  A minimum at or beyond 1e20 (no row may
    // score below the reference's starting value) or a non-finite one: every live row is scored.
    double bound = low * (1.0 + 2.1 * static_cast<double>(m) * 5.9604644775390625e-08 + 1e-9) + *st.scale * 1.4551915228366852e-11;
    if (!(low < 9e19)) bound = __builtin_inf();
    if (m > (1 << 18)) bound = __builtin_inf();
    for (int u = tid; u < n; u += 1024) {
        if (st.gone[u]) continue;
        bool in = st.irregular[u] != 0;
        if (!in) {
            const double s = all_of_them ? st.total[u] : st.total[u] - st.top[u];
            in = s <= bound;
        }
        if (in) st.contender[atomicAdd(&n_listed, 1)] = u;
    }
    __syncthreads();
    if (tid == 0) {
        *st.n_contenders = n_listed;
        *rescored += n_listed;
    }
}

// One wave per contender: defences.py:33-34 on the row as the reference sees it at pick t -- the sorted distances to the
// rows still there, the first m of them added left to right in fp32.
__global__ __launch_bounds__(256) void large_rescore_kernel(int t, int n, int users_count, int corrupted,
                                                            const float* __restrict__ sorted_val,
                                                            const uint32_t* __restrict__ sorted_idx, LargeState st,
                                                            const int32_t* __restrict__ status) {
    if (*status != 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    const int listed = *st.n_contenders;
    const int m = values_per_row(n, t, users_count, corrupted);
    for (int i = wave; i < listed; i += n_waves) {
        const int u = st.contender[i];
        const int64_t row = static_cast<int64_t>(u) * n;
        float s = 0.0f;
        int left = m;
        // 256 ranks per step, the next step's columns and values requested before this step's chain of additions: the walk is
        // two dependent loads per rank (the column, then whether that row is gone) and nothing else covers them
        constexpr int kAhead = 4;
        int col[kAhead];
        float val[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int r = 64 * k + lane;
            col[k] = r < n ? static_cast<int>(sorted_idx[row + r]) : u;
            val[k] = r < n ? sorted_val[row + r] : 0.0f;
        }
        for (int r0 = 0; r0 < n && left > 0; r0 += 64 * kAhead) {
            bool live[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) live[k] = col[k] != u && st.gone[col[k]] == 0;
            float v[kAhead];
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                v[k] = val[k];
                const int r = r0 + 64 * (kAhead + k) + lane;
                col[k] = r < n ? static_cast<int>(sorted_idx[row + r]) : u;
                val[k] = r < n ? sorted_val[row + r] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < kAhead; ++k) {
                const unsigned long long mask = __ballot(live[k]);
                if (mask == 0ull || left <= 0) continue;
                const int before = __popcll(mask & ((1ull << lane) - 1ull));
                const float x = live[k] && before < left ? v[k] : 0.0f;     // (s + 0.0 leaves s as it is: s is never -0.0, it starts at +0.0)
                const int last = 63 - __builtin_clzll(mask);
                for (int l = 0; l <= last; ++l) s = __fadd_rn(s, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)));
                left -= __popcll(mask);
            }
        }
        if (lane == 0) st.score[u] = s;
    }
}

}  // namespace

int segment_sort_u64(byz_ctx* ctx, unsigned long long* keys, int64_t n_segments, int64_t n_pad, hipStream_t stream) {
    BYZ_REQUIRE(keys && n_segments > 0 && n_pad >= 2 && (n_pad & (n_pad - 1)) == 0, "segment sort: bad arguments");
    (void)ctx;
    const int64_t chunks = n_pad < kChunk ? 1 : n_pad / kChunk;
    BYZ_REQUIRE(n_segments * chunks <= 0x7fffffff && ceil_div(n_segments * (n_pad / 2), 256) <= 0x7fffffff,
                "segment sort: too many keys for one launch");
    const unsigned local_grid = static_cast<unsigned>(n_segments * chunks);
    segment_sort_local_kernel<true><<<local_grid, kSortThreads, 0, stream>>>(keys, n_pad, static_cast<int>(chunks), 0);
    BYZ_TRY(check_launch("segment_sort_local_kernel"));
    const int64_t n_pairs = n_segments * (n_pad / 2);
    for (int64_t k = 2 * static_cast<int64_t>(kChunk); k <= n_pad; k <<= 1) {
        for (int64_t j = k / 2; j >= kChunk; j >>= 1) {
            segment_sort_global_kernel<<<static_cast<unsigned>(ceil_div(n_pairs, 256)), 256, 0, stream>>>(keys, n_pad, n_pairs, k, j);
            BYZ_TRY(check_launch("segment_sort_global_kernel"));
        }
        segment_sort_local_kernel<false><<<local_grid, kSortThreads, 0, stream>>>(keys, n_pad, static_cast<int>(chunks), k);
        BYZ_TRY(check_launch("segment_sort_local_kernel"));
    }
    return BYZ_OK;
}

size_t large_key_scratch_bytes() {
    if (const char* e = std::getenv("BYZ_LARGE_SCRATCH_MB")) {     // (tests: several batches at a small size)
        const long long mb = std::atoll(e);
        if (mb > 0) return static_cast<size_t>(mb) << 20;
    }
    return kKeyScratchBytes;
}

int launch_row_sort_large(byz_ctx* ctx, const float* dist, int64_t n, int64_t prefix_len, int64_t drop_count, bool want_tables,
                          hipStream_t stream) {
    BYZ_REQUIRE(dist && n > 0 && n < (int64_t{1} << 24), "row sort (large): bad arguments (n = %lld)", (long long)n);
    const int64_t n_pad = next_pow2(n < 2 ? 2 : n);
    int64_t rows_per_batch = static_cast<int64_t>(large_key_scratch_bytes() / (static_cast<size_t>(n_pad) * 8));
    if (rows_per_batch < 1) rows_per_batch = 1;
    if (rows_per_batch > n) rows_per_batch = n;
    if (rows_per_batch > 32768) rows_per_batch = 32768;     // (the keys kernel's grid.y)
    BYZ_TRY(ctx->large_keys.ensure(static_cast<size_t>(rows_per_batch) * n_pad * 8));
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(n) * sizeof(float)));
    BYZ_TRY(ctx->large_state.ensure(large_state_bytes(n)));
    if (want_tables) {
        BYZ_TRY(ctx->sorted_val.ensure(static_cast<size_t>(n) * n * sizeof(float) + 64));
        BYZ_TRY(ctx->large_idx.ensure(static_cast<size_t>(n) * n * sizeof(uint32_t)));
        BYZ_TRY(ctx->large_rank.ensure(static_cast<size_t>(n) * n * sizeof(uint32_t)));
    }
    const LargeState st = large_state(ctx->large_state.ptr, n);
    unsigned long long* keys = ctx->large_keys.as<unsigned long long>();
    KernelTimer timer(ctx, BYZ_K_ROW_SORT, stream);
    for (int64_t row0 = 0; row0 < n; row0 += rows_per_batch) {
        const int64_t rows = n - row0 < rows_per_batch ? n - row0 : rows_per_batch;
        large_row_keys_kernel<<<dim3(static_cast<unsigned>(ceil_div(n_pad, 256)), static_cast<unsigned>(rows)), 256, 0, stream>>>(
            dist, (int)n, n_pad, (int)row0, keys);
        BYZ_TRY(check_launch("large_row_keys_kernel"));
        BYZ_TRY(segment_sort_u64(ctx, keys, rows, n_pad, stream));
        if (want_tables)
            large_row_tables_kernel<true><<<static_cast<unsigned>(rows), 256, 0, stream>>>(
                keys, (int)n, n_pad, (int)row0, (int)prefix_len, (int)drop_count, ctx->scores.as<float>(), ctx->sorted_val.as<float>(),
                ctx->large_idx.as<uint32_t>(), ctx->large_rank.as<uint32_t>(), st);
        else
            large_row_tables_kernel<false><<<static_cast<unsigned>(rows), 256, 0, stream>>>(
                keys, (int)n, n_pad, (int)row0, (int)prefix_len, (int)drop_count, ctx->scores.as<float>(), nullptr, nullptr, nullptr, st);
        BYZ_TRY(check_launch("large_row_tables_kernel"));
    }
    return BYZ_OK;
}

int launch_bulyan_loop_large(byz_ctx* ctx, const float* dist, int64_t n, int64_t theta, int64_t drop_count, int64_t users_count,
                             int64_t corrupted, int32_t* selection_dev, int32_t* status_dev, hipStream_t stream) {
    BYZ_REQUIRE(dist && selection_dev && status_dev && n > 0 && theta >= 0 && theta <= n,
                "bulyan loop (large): bad arguments (n=%lld theta=%lld)", (long long)n, (long long)theta);
    BYZ_REQUIRE(ctx->large_state.bytes >= large_state_bytes(n) && ctx->large_idx.ptr && ctx->large_rank.ptr,
                "bulyan loop (large): the row sort has not run");
    const LargeState st = large_state(ctx->large_state.ptr, n);
    BYZ_HIP(hipMemsetAsync(status_dev, 0, 3 * sizeof(int32_t), stream));   // status, rows re-scored, (unused)
    BYZ_HIP(hipMemsetAsync(st.n_contenders, 0, 2 * sizeof(int32_t), stream));
    KernelTimer timer(ctx, BYZ_K_BULYAN_LOOP, stream);
    int64_t waves = static_cast<int64_t>(ctx->num_cus) * 16;
    const unsigned rescore_grid = static_cast<unsigned>(waves / 4);
    for (int64_t t = 0; t <= theta; ++t) {
        large_pick_kernel<<<1, 1024, 0, stream>>>((int)t, (int)n, (int)theta, (int)drop_count, (int)users_count, (int)corrupted,
                                                  ctx->sorted_val.as<float>(), ctx->large_idx.as<uint32_t>(),
                                                  ctx->large_rank.as<uint32_t>(), st, selection_dev, status_dev, status_dev + 1);
        if (t == theta) break;
        large_rescore_kernel<<<rescore_grid, 256, 0, stream>>>((int)t, (int)n, (int)users_count, (int)corrupted,
                                                               ctx->sorted_val.as<float>(), ctx->large_idx.as<uint32_t>(), st, status_dev);
    }
    return check_launch("large_pick_kernel / large_rescore_kernel");
}

}  // namespace byz
