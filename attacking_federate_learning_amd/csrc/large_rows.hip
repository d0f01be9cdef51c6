// More rows than the LDS-resident kernels hold (select.hip, trimmed_mean.hip: 16,384).  The reference has no limit
// (defences.py:23-70 are loops over Python lists); no BASELINE configuration goes beyond N = 10,000, so this file is first of
// all about BEING THERE with the reference's results: a textbook sort on global memory, 32-bit indices throughout -- and then
// about not being slow (EXPERIMENTS.md L1: the Bulyan loop went from 7.1 to 0.9 s at N = 20,000).
//
//   segment_sort_u64            many independent arrays of 64-bit keys, each a power of two long, sorted ascending: a bitonic
//                               network whose merges of up to 4096 keys run in LDS and whose longer strides are one launch each
//   launch_row_sort_large       defences.py:31-34: every row of the N x N distance matrix sorted (the same keys as select.hip:
//                               order-preserving float bits << 32 | column, the self entry last, a NaN behind +inf), the Krum
//                               score as the SEQUENTIAL fp32 sum of the first `prefix_len` sorted values; for Bulyan the sorted
//                               values, the sorted columns, the rank of every column and two fp64 sums per row
//   launch_bulyan_loop_large    defences.py:59-68, three launches per batch of 16 picks.  A row's exact score -- the sum of all its live
//                               distances minus the sum of the `drop` largest, both carried in fp64 and updated in O(1) when a
//                               row leaves -- bounds the reference's sequential fp32 sum from both sides ((1 +- u)^(m-1),
//                               u = 2^-24); every live row whose lower bound does not exceed the smallest upper bound is a
//                               CONTENDER and is scored again exactly the reference's way (a walk along its sorted row that
//                               skips the rows already picked and adds the first m live values left to right in fp32); the
//                               reference's strict '<' in its visit order 1, 0, 2, ... then decides among the contenders.  A row
//                               outside the band cannot win or tie, so the selection is the reference's, pick for pick.  Rows
//                               with a negative or non-finite entry (an arbitrary caller-supplied matrix) always contend.
#include "common.hpp"

#include <hip/hip_cooperative_groups.h>

#include <cstdlib>

namespace byz {
namespace {

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

// FIRST: every merge k = 2 .. len of one chunk; else the levels j = len / 2 .. 1 of the merge of size k_merge.  A chunk is 32 KiB of
// keys: 4096 of 64 bits (value | column: the selection), 8192 of 32 bits (values alone: the trimmed mean).
template <typename K>
constexpr int chunk_keys() { return static_cast<int>(32768 / sizeof(K)); }

template <typename K, bool FIRST>
__global__ __launch_bounds__(kSortThreads) void segment_sort_local_kernel(K* __restrict__ keys, int64_t n_pad, int chunks_per_seg,
                                                                          int64_t k_merge) {
    constexpr int CH = chunk_keys<K>();
    __shared__ K lds[CH];
    const int64_t seg = blockIdx.x / chunks_per_seg;
    const int64_t ch = blockIdx.x % chunks_per_seg;
    const int len = n_pad < CH ? static_cast<int>(n_pad) : CH;
    const int64_t i0 = ch * CH;                       // position of lds[0] inside its segment
    K* const base = keys + seg * n_pad + i0;
    const int tid = threadIdx.x;
    for (int i = tid; i < len; i += kSortThreads) lds[i] = base[i];
    __syncthreads();
    for (int64_t k = FIRST ? 2 : k_merge; k <= (FIRST ? static_cast<int64_t>(len) : k_merge); k <<= 1) {
        const int j_top = k / 2 < len / 2 ? static_cast<int>(k / 2) : len / 2;
        for (int j = j_top; j > 0; j >>= 1) {
            for (int p = tid; p < len / 2; p += kSortThreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int q = i | j;
                const K a = lds[i], b = lds[q];
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

// one level (stride j >= a chunk) of the merge of size k, every segment at once
template <typename K>
__global__ __launch_bounds__(256) void segment_sort_global_kernel(K* __restrict__ keys, int64_t n_pad, int64_t n_pairs, int64_t k,
                                                                  int64_t j) {
    const int64_t g = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (g >= n_pairs) return;
    const int64_t half = n_pad >> 1;
    const int64_t seg = g / half, p = g % half;
    const int64_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
    const int64_t q = i | j;
    K* const base = keys + seg * n_pad;
    const K a = base[i], b = base[q];
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

// Krum alone needs no columns: the sorted VALUES are all a score is made of (defences.py:33-34), so its keys are 32 bits (half the
// bytes through the sort, 8192 keys per LDS chunk)
__global__ __launch_bounds__(256) void large_row_keys32_kernel(const float* __restrict__ dist, int n, int64_t n_pad, int row0,
                                                               uint32_t* __restrict__ keys) {
    const int u = row0 + blockIdx.y;
    const int64_t c = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (c >= n_pad) return;
    uint32_t key = 0xffffffffu;      // padding, and the self entry: behind every real one (the prefix never reaches rank n - 1)
    if (c < n && c != u) {
        const float d = dist[static_cast<int64_t>(u) * n + c];
        key = d != d ? 0xfffffffeu : ordered_bits(d);
    }
    keys[static_cast<int64_t>(blockIdx.y) * n_pad + c] = key;
}

// one wave per sorted row: Python's sum() over np.float32 scalars (defences.py:34): 0 + x0, then one fp32 addition per value
__global__ __launch_bounds__(64) void large_row_scores32_kernel(const uint32_t* __restrict__ keys, int64_t n_pad, int row0,
                                                                int prefix_len, float* __restrict__ scores) {
    const int lane = threadIdx.x;
    const uint32_t* const row = keys + static_cast<int64_t>(blockIdx.x) * n_pad;
    float s = 0.0f;
    for (int r0 = 0; r0 < prefix_len; r0 += 64) {
        const int r = r0 + lane;
        const float v = r < prefix_len ? from_ordered_bits(row[r]) : 0.0f;
        const int cnt = prefix_len - r0 < 64 ? prefix_len - r0 : 64;
        for (int i = 0; i < cnt; ++i) s = __fadd_rn(s, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)));
    }
    if (lane == 0) scores[row0 + blockIdx.x] = s;
}

// the state of the Bulyan loop (ctx->large_state): doubles first, then the 32-bit words
constexpr int kBatchMax = 32;          // picks decided on the exact scores before their contenders are scored together
constexpr int kNever = 0x7fffffff;     // gone_at of a row that is still there
struct LargeState {
    double* total;        // [n] sum of the row's live distances (regular rows)
    double* top;          // [n] sum of its `drop` largest live distances
    double* total_kept;   // [n] the three of them as they were when the batch began (what a roll-back starts from)
    double* top_kept;     // [n]
    double* exact;        // [n] the exact score at the pick being decided (-inf: the row always contends, NaN: it is gone)
    double* scale;        // [2] the largest first total: what the absolute rounding slack of the running sums is taken from
    int32_t* top_first;   // [n] the lowest rank that belongs to the `drop` largest live entries
    int32_t* first_kept;  // [n]
    int32_t* irregular;   // [n] the row holds a negative or non-finite distance: it is scored the reference's way at every pick
    int32_t* gone_at;     // [n] the pick that took the row (kNever: still there); live in the state of pick t <=> gone_at >= t
    int32_t* leader;      // [n] by class (= its smallest row): the member the reference visits first among those still there (-1: none)
    int32_t* leader_kept; // [n]
    int32_t* next_twin;   // [n] the next member of the row's class in visit order (-1: the last one)
    const int32_t* cls;   // [n] the row's twin class (select.hip's twin_class kernels; ctx->twin_class)
    int32_t* pair_row;    // [kBatchMax n] the contenders of the batch's picks, pick after pick
    float* pair_score;    // [kBatchMax n] their scores in the state of their pick
    int32_t* pair_begin;  // [kBatchMax + 1] where a pick's contenders begin
    int32_t* guess;       // [kBatchMax] the row with the smallest exact score at that pick (-1: none)
    int32_t* words;       // [4] the next pick, the picks of the current batch, (unused)
};
__host__ __device__ inline size_t large_state_bytes(int64_t n) {
    return static_cast<size_t>(5 * n + 2) * sizeof(double) + static_cast<size_t>(7 * n + 2 * kBatchMax * n + 2 * kBatchMax + 8) * sizeof(int32_t) + 64;
}
__host__ __device__ inline LargeState large_state(void* p, int64_t n) {
    LargeState s;
    s.total = static_cast<double*>(p);
    s.top = s.total + n;
    s.total_kept = s.top + n;
    s.top_kept = s.total_kept + n;
    s.exact = s.top_kept + n;
    s.scale = s.exact + n;
    s.top_first = reinterpret_cast<int32_t*>(s.scale + 2);
    s.first_kept = s.top_first + n;
    s.irregular = s.first_kept + n;
    s.gone_at = s.irregular + n;
    s.leader = s.gone_at + n;
    s.leader_kept = s.leader + n;
    s.next_twin = s.leader_kept + n;
    s.cls = nullptr;
    s.pair_row = s.next_twin + n;
    s.pair_score = reinterpret_cast<float*>(s.pair_row + static_cast<int64_t>(kBatchMax) * n);
    s.pair_begin = reinterpret_cast<int32_t*>(s.pair_score + static_cast<int64_t>(kBatchMax) * n);
    s.guess = s.pair_begin + kBatchMax + 1;
    s.words = s.guess + kBatchMax;
    return s;
}

// One workgroup per sorted row (Bulyan; Krum alone takes the 32-bit kernels above): the Krum score, the tables and the row's state.
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
    {
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
            st.gone_at[u] = kNever;
        }
    }
}

// out[c][r] = in[r][c] of an n x n matrix of 32-bit words: 64 x 64 tiles through LDS, both sides in contiguous runs
__global__ __launch_bounds__(256) void transpose32_kernel(const uint32_t* __restrict__ in, int n, uint32_t* __restrict__ out) {
    __shared__ uint32_t tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = r < n && c < n ? in[static_cast<int64_t>(r) * n + c] : 0u;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < n && r < n) out[static_cast<int64_t>(c) * n + r] = tile[tx][i];
    }
}

// ---- the Bulyan loop ----------------------------------------------------------------------------------------------------
// Three launches per BATCH of picks (first form: two per pick, 0.57 ms a pick at N = 20,000 -- the walk of one contender is ~15,000
// dependent additions, and nothing ran beside it):
//   large_decide_kernel   one workgroup: up to `batch` picks decided on the EXACT scores (the row with the smallest one, ties by
//                         visit order, leaves and every row's sums follow), the contenders of every pick listed in the state of
//                         that pick;
//   large_rescore_kernel  every (pick, contender) pair scored the reference's way, one wave each, in the state of its pick (a
//                         row is there at pick t while gone_at >= t);
//   large_settle_kernel   one workgroup: pick after pick the reference's comparison loop over the contenders' fp32 scores; where
//                         its winner is not the guess, the batch is cut there: the sums go back to what they were when the batch
//                         began, the picks that stood and the true winner are replayed, and the next batch starts behind it.
// The same selection pick for pick whatever the batch length (BYZ_LARGE_BATCH, 1 .. 32).
struct Pick {
    float score;
    int pos;
    int row;
};
__device__ __forceinline__ bool better(const Pick& a, const Pick& b) {
    return a.score < b.score || (a.score == b.score && a.pos < b.pos);   // strict '<'; an equal score keeps the earlier visitor
}
struct Guess {
    double score;
    int pos;
    int row;
};
__device__ __forceinline__ bool better(const Guess& a, const Guess& b) { return a.score < b.score || (a.score == b.score && a.pos < b.pos); }

// how many live values pick t adds per row: users_count - len(selection_set) - corrupted_count of the n - t - 1 that are left
// (defences.py:26, 34, 61)
__device__ __forceinline__ int values_per_row(int n, int t, int users_count, int corrupted) {
    const int want = users_count - t - corrupted, left = n - t - 1;
    return want < left ? (want > 0 ? want : 0) : left;
}

// Row w leaves at pick t (gone_at[w] == t already): every row that stays takes d(u, w) out of its sums.  The whole workgroup.
// (rank_t[w][u] = the rank of column w in row u and dist_t[w][u] = d(u, w): the winner's ROWS of the transposed tables, read
// contiguously; out of the rows' own tables they were two scattered lines per row and pick -- 5 MB through one CU at 20,000 rows)
__device__ __forceinline__ void remove_from_rows(int w, int t, int n, int drop, const float* __restrict__ sorted_val,
                                                 const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ rank_t,
                                                 const float* __restrict__ dist_t, const LargeState& st) {
    constexpr int kRows = 4;     // rows per thread and step: their dependent loads (the rank, then the value) side by side
    for (int base = threadIdx.x; base < n; base += kRows * 1024) {
        bool stays[kRows];
        int r[kRows];
        float v[kRows];
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int u = base + j * 1024;
            stays[j] = u < n && st.gone_at[u] > t && st.irregular[u] == 0;
            r[j] = stays[j] ? static_cast<int>(rank_t[static_cast<int64_t>(w) * n + u]) : 0;
            v[j] = stays[j] ? dist_t[static_cast<int64_t>(w) * n + u] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            if (!stays[j]) continue;
            const int u = base + j * 1024;
            const int64_t row = static_cast<int64_t>(u) * n;
            const double d = static_cast<double>(v[j]);
            st.total[u] -= d;
            if (drop > 0 && r[j] >= st.top_first[u]) {
                // the winner was one of the `drop` largest: the next live entry below them takes its place
                int p = st.top_first[u] - 1;
                while (p >= 0 && st.gone_at[sorted_idx[row + p]] <= t) --p;
                st.top[u] += (p >= 0 ? static_cast<double>(sorted_val[row + p]) : 0.0) - d;
                st.top_first[u] = p;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void large_decide_kernel(int n, int theta, int drop, int users_count, int corrupted, int batch,
                                                            const float* __restrict__ sorted_val,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const uint32_t* __restrict__ rank_t,
                                                            const float* __restrict__ dist_t, LargeState st,
                                                            const int32_t* __restrict__ status, int32_t* __restrict__ rescored) {
    __shared__ Guess best[16];
    __shared__ int n_listed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = st.words[0];
    if (*status != 0 || t0 >= theta) {
        if (tid == 0) st.words[1] = 0;
        return;
    }
    if (t0 == 0) {
        double big = 0.0;
        for (int u = tid; u < n; u += 1024) big = st.irregular[u] ? big : (st.total[u] > big ? st.total[u] : big);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const double o = __shfl_xor(big, m, 64);
            big = o > big ? o : big;
        }
        if (lane == 0) best[wave].score = big;
        __syncthreads();
        if (tid == 0) {
            double all = 0.0;
            for (int w = 0; w < 16; ++w) all = best[w].score > all ? best[w].score : all;
            *st.scale = all;
        }
        __syncthreads();
    }
    if (tid == 0) {
        n_listed = 0;
        st.pair_begin[0] = 0;
    }
    const double slack = *st.scale * 1.4551915228366852e-11;     // 2^17 roundings of 2^-53 of the largest first total
    int done = 0;
    int w_prev = -1;      // the row the pick before took: it leaves every row's sums in the same pass that forms the next scores
    for (int k = 0; k < batch && t0 + k < theta; ++k) {
        const int t = t0 + k;
        const int m = values_per_row(n, t, users_count, corrupted);
        const bool all_of_them = users_count - t - corrupted >= n - t - 1;     // nothing is dropped: the score is the whole total
        Guess mine{__builtin_inf(), 0x7fffffff, -1};
        // ONE pass over the rows per pick (the first form made three: the guess, the list, the removal -- 0.2 ms a pick at 20,000
        // rows): the row the pick before took leaves the sums, the exact score is formed and kept for the list below
        constexpr int kRows = 4;     // rows per thread and step: their dependent loads (the rank, then the value) side by side
        for (int base = tid; base < n; base += kRows * 1024) {
            bool there[kRows], regular[kRows];
            int r[kRows];
            float v[kRows];
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int u = base + j * 1024;
                there[j] = u < n && st.gone_at[u] >= t;
                regular[j] = there[j] && st.irregular[u] == 0;
                r[j] = regular[j] && w_prev >= 0 ? static_cast<int>(rank_t[static_cast<int64_t>(w_prev) * n + u]) : 0;
                v[j] = regular[j] && w_prev >= 0 ? dist_t[static_cast<int64_t>(w_prev) * n + u] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < kRows; ++j) {
                const int u = base + j * 1024;
                if (u >= n) continue;
                if (k == 0) st.leader_kept[u] = st.leader[u];
                if (!regular[j]) {
                    st.exact[u] = there[j] ? -__builtin_inf() : __builtin_nan("");
                    continue;
                }
                double tot = st.total[u], top = st.top[u];
                if (k == 0) {     // what a roll-back starts from
                    st.total_kept[u] = tot;
                    st.top_kept[u] = top;
                    st.first_kept[u] = st.top_first[u];
                }
                if (w_prev >= 0) {
                    const int64_t row = static_cast<int64_t>(u) * n;
                    const double d = static_cast<double>(v[j]);
                    tot -= d;
                    st.total[u] = tot;
                    if (drop > 0 && r[j] >= st.top_first[u]) {
                        // the winner was one of the `drop` largest: the next live entry below them takes its place
                        int p = st.top_first[u] - 1;
                        while (p >= 0 && st.gone_at[sorted_idx[row + p]] < t) --p;
                        top += (p >= 0 ? static_cast<double>(sorted_val[row + p]) : 0.0) - d;
                        st.top[u] = top;
                        st.top_first[u] = p;
                    }
                }
                const Guess o{all_of_them ? tot : tot - top, visit_position(u), u};
                st.exact[u] = o.score;
                if (better(o, mine)) mine = o;
            }
        }
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) {
            const Guess o{__shfl_xor(mine.score, x, 64), __shfl_xor(mine.pos, x, 64), __shfl_xor(mine.row, x, 64)};
            if (better(o, mine)) mine = o;
        }
        __syncthreads();      // (best[] may still be read from the pick before)
        if (lane == 0) best[wave] = mine;
        __syncthreads();
        mine = best[0];
        for (int w = 1; w < 16; ++w)
            if (better(best[w], mine)) mine = best[w];
        const double low = mine.score;
        // fl(sum) of m non-negative terms added one by one lies in [s (1 - u)^(m-1), s (1 + u)^(m-1)], u = 2^-24: a row can reach
        // the smallest sum only from below s_min ((1 + u) / (1 - u))^(m-1) <= s_min (1 + 2.1 m u) for m u <= 2^-6 (scripts/proto/
        // band_loop.py, tests/test_proto_rules.py).  A minimum at or beyond 1e20 (no row may score below the reference's starting
        // value), a non-finite one, or more values than the bound was derived for: every live row is scored.
        double bound = low * (1.0 + 2.1 * static_cast<double>(m) * 5.9604644775390625e-08 + 1e-9) + slack;
        if (!(low < 9e19) || m > (1 << 18)) bound = __builtin_inf();
        for (int u = tid; u < n; u += 1024) {
            const double e = st.exact[u];
            // (false for the NaN of a row that is gone; of a twin class only the member the reference visits first; -inf: always)
            if (e <= bound && (e == -__builtin_inf() || st.leader[st.cls[u]] == u)) st.pair_row[atomicAdd(&n_listed, 1)] = u;
        }
        __syncthreads();
        const int w = mine.row;
        if (tid == 0) {
            st.pair_begin[k + 1] = n_listed;
            st.guess[k] = w;
            if (w >= 0) {
                st.gone_at[w] = t;
                if (st.leader[st.cls[w]] == w) st.leader[st.cls[w]] = st.next_twin[w];
            }
        }
        done = k + 1;
        w_prev = w;
        __threadfence_block();
        __syncthreads();
        if (w < 0) break;      // no regular row is left: the contenders' scores decide this pick, and the batch ends with it
    }
    if (w_prev >= 0) {         // the batch's last guess leaves the sums too: the state the next batch starts from
        remove_from_rows(w_prev, t0 + done - 1, n, drop, sorted_val, sorted_idx, rank_t, dist_t, st);
        __syncthreads();
    }
    if (tid == 0) {
        st.words[1] = done;
        const int pairs = n_listed;
        *rescored = *rescored > 0x7fffffff - pairs ? 0x7fffffff : *rescored + pairs;
    }
}

// Twin classes (select.hip: rows that are bitwise equal but for their own and each other's zero -- the attack's identical clients,
// malicious.py:26-27): their live distance multisets stay equal through every removal, so their fp32 scores are one number and the
// reference's strict '<' keeps the member it visits first.  Only that LEADER is ever listed and scored; when it leaves, the next
// member in visit order leads.  (4,800 twins among 20,000 rows were 4,800 walks per pick.)
__global__ __launch_bounds__(256) void large_twin_leader_kernel(int n, LargeState st) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    // (before the loop: leader_kept holds the smallest visit POSITION of every class, 0x7fffffff where a row roots none; first_kept
    // the classes' sizes)
    atomicMin(&st.leader_kept[st.cls[u]], visit_position(u));
    atomicAdd(&st.first_kept[st.cls[u]], 1);
}
__global__ __launch_bounds__(256) void large_twin_links_kernel(int n, LargeState st) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const int c = st.cls[u];
    const int first = st.leader_kept[u];
    st.leader[u] = first == 0x7fffffff ? -1 : visit_position(first);      // (the visit order 1, 0, 2, 3, ... is an involution)
    int next = -1;
    if (st.first_kept[c] > 1) {
        for (int p = visit_position(u) + 1; p < n; ++p) {
            const int v = visit_position(p);
            if (st.cls[v] == c) {
                next = v;
                break;
            }
        }
    }
    st.next_twin[u] = next;
}

// The largest first total (the absolute slack of the running fp64 sums is taken from it), once per loop.
__global__ __launch_bounds__(1024) void large_scale_kernel(int n, LargeState st) {
    __shared__ double part[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double big = 0.0;
    for (int u = tid; u < n; u += 1024) big = st.irregular[u] ? big : (st.total[u] > big ? st.total[u] : big);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        const double o = __shfl_xor(big, m, 64);
        big = o > big ? o : big;
    }
    if (lane == 0) part[wave] = big;
    __syncthreads();
    if (tid == 0) {
        double all = 0.0;
        for (int w = 0; w < 16; ++w) all = part[w] > all ? part[w] : all;
        *st.scale = all;
    }
}

// large_decide_kernel on MANY workgroups (a cooperative launch: all of them resident, two grid-wide barriers per pick): the rows
// spread over the whole chip instead of one compute unit -- what a pick cost in the one-workgroup form was that one unit's pass
// over every row's sums.  Per pick: every workgroup's best exact score -> barrier -> everybody forms the same minimum and bound,
// lists its rows inside the band (one reservation of the global list per workgroup), thread 0 books the guess -> barrier.
constexpr int kGridBlock = 256;
constexpr int kGridMaxBlocks = 256;
__global__ __launch_bounds__(kGridBlock) void large_decide_grid_kernel(int n, int theta, int drop, int users_count, int corrupted,
                                                                       int batch, const float* __restrict__ sorted_val,
                                                                       const uint32_t* __restrict__ sorted_idx,
                                                                       const uint32_t* __restrict__ rank_t,
                                                                       const float* __restrict__ dist_t, LargeState st,
                                                                       const int32_t* __restrict__ status, int32_t* __restrict__ rescored,
                                                                       Guess* __restrict__ wg_best, int32_t* __restrict__ n_listed) {
    namespace cg = cooperative_groups;
    cg::grid_group grid = cg::this_grid();
    extern __shared__ int32_t lds_list[];      // the rows of this workgroup inside the band (at most its share of the rows)
    __shared__ Guess best[kGridBlock / 64];
    __shared__ int lds_count, lds_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gtid = blockIdx.x * kGridBlock + tid, gthreads = gridDim.x * kGridBlock;
    const int t0 = st.words[0];
    if (*status != 0 || t0 >= theta) {          // (the same for every workgroup: nobody reaches a barrier)
        if (gtid == 0) st.words[1] = 0;
        return;
    }
    if (gtid == 0) {
        *n_listed = 0;
        st.pair_begin[0] = 0;
    }
    const double slack = *st.scale * 1.4551915228366852e-11;     // 2^17 roundings of 2^-53 of the largest first total
    int done = 0;
    int w_prev = -1;
    for (int k = 0; k < batch && t0 + k < theta; ++k) {
        const int t = t0 + k;
        const int m = values_per_row(n, t, users_count, corrupted);
        const bool all_of_them = users_count - t - corrupted >= n - t - 1;
        Guess mine{__builtin_inf(), 0x7fffffff, -1};
        for (int u = gtid; u < n; u += gthreads) {
            const bool there = st.gone_at[u] >= t;
            if (k == 0) st.leader_kept[u] = st.leader[u];
            if (!there || st.irregular[u] != 0) {
                st.exact[u] = there ? -__builtin_inf() : __builtin_nan("");
                continue;
            }
            double tot = st.total[u], top = st.top[u];
            if (k == 0) {     // what a roll-back starts from
                st.total_kept[u] = tot;
                st.top_kept[u] = top;
                st.first_kept[u] = st.top_first[u];
            }
            if (w_prev >= 0) {
                const int r = static_cast<int>(rank_t[static_cast<int64_t>(w_prev) * n + u]);
                const double d = static_cast<double>(dist_t[static_cast<int64_t>(w_prev) * n + u]);
                tot -= d;
                st.total[u] = tot;
                if (drop > 0 && r >= st.top_first[u]) {
                    const int64_t row = static_cast<int64_t>(u) * n;
                    int p = st.top_first[u] - 1;
                    while (p >= 0 && st.gone_at[sorted_idx[row + p]] < t) --p;
                    top += (p >= 0 ? static_cast<double>(sorted_val[row + p]) : 0.0) - d;
                    st.top[u] = top;
                    st.top_first[u] = p;
                }
            }
            const Guess o{all_of_them ? tot : tot - top, visit_position(u), u};
            st.exact[u] = o.score;
            if (better(o, mine)) mine = o;
        }
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) {
            const Guess o{__shfl_xor(mine.score, x, 64), __shfl_xor(mine.pos, x, 64), __shfl_xor(mine.row, x, 64)};
            if (better(o, mine)) mine = o;
        }
        if (lane == 0) best[wave] = mine;
        if (tid == 0) lds_count = 0;
        __syncthreads();
        if (tid == 0) {
            Guess b = best[0];
            for (int w = 1; w < kGridBlock / 64; ++w)
                if (better(best[w], b)) b = best[w];
            wg_best[blockIdx.x] = b;
        }
        grid.sync();
        mine = Guess{__builtin_inf(), 0x7fffffff, -1};
        for (int b = lane; b < static_cast<int>(gridDim.x); b += 64) {
            const Guess o = wg_best[b];
            if (better(o, mine)) mine = o;
        }
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) {
            const Guess o{__shfl_xor(mine.score, x, 64), __shfl_xor(mine.pos, x, 64), __shfl_xor(mine.row, x, 64)};
            if (better(o, mine)) mine = o;
        }
        const double low = mine.score;
        double bound = low * (1.0 + 2.1 * static_cast<double>(m) * 5.9604644775390625e-08 + 1e-9) + slack;      // (large_decide_kernel)
        if (!(low < 9e19) || m > (1 << 18)) bound = __builtin_inf();
        for (int u = gtid; u < n; u += gthreads) {
            const double e = st.exact[u];
            // (false for the NaN of a row that is gone; of a twin class only the member the reference visits first; -inf: always)
            if (e <= bound && (e == -__builtin_inf() || st.leader[st.cls[u]] == u)) lds_list[atomicAdd(&lds_count, 1)] = u;
        }
        __syncthreads();
        if (tid == 0) lds_base = lds_count > 0 ? atomicAdd(n_listed, lds_count) : 0;
        __syncthreads();
        for (int i = tid; i < lds_count; i += kGridBlock) st.pair_row[lds_base + i] = lds_list[i];
        const int w = mine.row;
        if (gtid == 0) {
            st.guess[k] = w;
            if (w >= 0) st.gone_at[w] = t;
        }
        done = k + 1;
        w_prev = w;
        grid.sync();
        if (gtid == 0) {
            st.pair_begin[k + 1] = *n_listed;
            // (only now: other workgroups were still listing THIS pick by its leaders; the next pick's list is behind the next barrier)
            if (w >= 0 && st.leader[st.cls[w]] == w) st.leader[st.cls[w]] = st.next_twin[w];
        }
        if (w < 0) break;
    }
    if (w_prev >= 0) {         // the batch's last guess leaves the sums too: the state the next batch starts from
        const int t = t0 + done - 1;
        for (int u = gtid; u < n; u += gthreads) {
            if (st.gone_at[u] <= t || st.irregular[u] != 0) continue;
            const int r = static_cast<int>(rank_t[static_cast<int64_t>(w_prev) * n + u]);
            const double d = static_cast<double>(dist_t[static_cast<int64_t>(w_prev) * n + u]);
            st.total[u] -= d;
            if (drop > 0 && r >= st.top_first[u]) {
                const int64_t row = static_cast<int64_t>(u) * n;
                int p = st.top_first[u] - 1;
                while (p >= 0 && st.gone_at[sorted_idx[row + p]] <= t) --p;
                st.top[u] += (p >= 0 ? static_cast<double>(sorted_val[row + p]) : 0.0) - d;
                st.top_first[u] = p;
            }
        }
    }
    if (gtid == 0) {
        st.words[1] = done;
        const int pairs = *n_listed;
        *rescored = *rescored > 0x7fffffff - pairs ? 0x7fffffff : *rescored + pairs;
    }
}

// One wave per (pick, contender): defences.py:33-34 on the row as the reference sees it at that pick -- the sorted distances to
// the rows still there, the first m of them added left to right in fp32.
__global__ __launch_bounds__(256) void large_rescore_kernel(int n, int users_count, int corrupted, const float* __restrict__ sorted_val,
                                                            const uint32_t* __restrict__ sorted_idx, LargeState st,
                                                            const int32_t* __restrict__ status) {
    const int picks = st.words[1];
    if (*status != 0 || picks == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
    const int t0 = st.words[0];
    const int listed = st.pair_begin[picks];
    for (int i = wave; i < listed; i += n_waves) {
        int k = 0;
        while (k + 1 < picks && st.pair_begin[k + 1] <= i) ++k;
        const int t = t0 + k;
        const int m = values_per_row(n, t, users_count, corrupted);
        const int u = st.pair_row[i];
        const int64_t row = static_cast<int64_t>(u) * n;
        float s = 0.0f;
        int left = m;
        // 256 ranks per step, the next step's columns and values requested before this step's chain of additions: the walk is
        // two dependent loads per rank (the column, then when that row left) and nothing else covers them
        constexpr int kAhead = 4;
        int col[kAhead];
        float val[kAhead];
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
            const int r = 64 * a + lane;
            col[a] = r < n ? static_cast<int>(sorted_idx[row + r]) : u;
            val[a] = r < n ? sorted_val[row + r] : 0.0f;
        }
        for (int r0 = 0; r0 < n && left > 0; r0 += 64 * kAhead) {
            bool live[kAhead];
#pragma unroll
            for (int a = 0; a < kAhead; ++a) live[a] = col[a] != u && st.gone_at[col[a]] >= t;
            float v[kAhead];
#pragma unroll
            for (int a = 0; a < kAhead; ++a) {
                v[a] = val[a];
                const int r = r0 + 64 * (kAhead + a) + lane;
                col[a] = r < n ? static_cast<int>(sorted_idx[row + r]) : u;
                val[a] = r < n ? sorted_val[row + r] : 0.0f;
            }
#pragma unroll
            for (int a = 0; a < kAhead; ++a) {
                const unsigned long long mask = __ballot(live[a]);
                if (mask == 0ull || left <= 0) continue;
                const int before = __popcll(mask & ((1ull << lane) - 1ull));
                const float x = live[a] && before < left ? v[a] : 0.0f;     // (s + 0.0 leaves s as it is: s is never -0.0, it starts at +0.0)
                const int last = 63 - __builtin_clzll(mask);
                for (int l = 0; l <= last; ++l) s = __fadd_rn(s, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)));
                left -= __popcll(mask);
            }
        }
        if (lane == 0) st.pair_score[i] = s;
    }
}

__global__ __launch_bounds__(1024) void large_settle_kernel(int n, int theta, int drop, const float* __restrict__ sorted_val,
                                                            const uint32_t* __restrict__ sorted_idx,
                                                            const uint32_t* __restrict__ rank_t,
                                                            const float* __restrict__ dist_t, LargeState st,
                                                            int32_t* __restrict__ selection, int32_t* __restrict__ status) {
    __shared__ Pick picks_sh[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int picks = st.words[1];
    if (*status != 0 || picks == 0) return;
    const int t0 = st.words[0];
    for (int k = 0; k < picks; ++k) {
        // defences.py:27-37 over the rows that can win: strict '<' from 1e20 in the visit order 1, 0, 2, ...
        Pick mine{kKrumInit, 0x7fffffff, -1};
        for (int i = st.pair_begin[k] + tid; i < st.pair_begin[k + 1]; i += 1024) {
            const float s = st.pair_score[i];
            if (s < kKrumInit) {   // false for NaN, as in the reference's comparison
                const int u = st.pair_row[i];
                const Pick o{s, visit_position(u), u};
                if (better(o, mine)) mine = o;
            }
        }
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) {
            const Pick o{__shfl_xor(mine.score, x, 64), __shfl_xor(mine.pos, x, 64), __shfl_xor(mine.row, x, 64)};
            if (better(o, mine)) mine = o;
        }
        __syncthreads();
        if (lane == 0) picks_sh[wave] = mine;
        __syncthreads();
        mine = picks_sh[0];
        for (int w = 1; w < 16; ++w)
            if (better(picks_sh[w], mine)) mine = picks_sh[w];
        const int winner = mine.row;
        if (winner == st.guess[k] && winner >= 0) {
            if (tid == 0) selection[t0 + k] = winner;
            continue;
        }
        // The guess did not stand.  The sums go back to the start of the batch; the rows the batch took from pick k on are there
        // again; the picks that stood, and this pick's true winner, are replayed.  (No winner at all -- minimal_error_index stays
        // -1, the reference's distances.pop(-1) raises KeyError -- ends the loop: status 1.)
        __syncthreads();
        for (int u = tid; u < n; u += 1024) {
            st.leader[u] = st.leader_kept[u];
            if (st.gone_at[u] < t0 || st.irregular[u]) continue;      // (gone before the batch / never summed: nothing was kept)
            st.total[u] = st.total_kept[u];
            st.top[u] = st.top_kept[u];
            st.top_first[u] = st.first_kept[u];
            if (st.gone_at[u] >= t0 + k && st.gone_at[u] != kNever) st.gone_at[u] = kNever;
        }
        __syncthreads();
        if (winner < 0) {
            if (tid == 0) {
                *status = 1;
                st.words[0] = t0 + k;
            }
            return;
        }
        if (tid == 0) {
            selection[t0 + k] = winner;
            st.gone_at[winner] = t0 + k;
        }
        __threadfence_block();
        __syncthreads();
        for (int j = 0; j <= k; ++j) {
            const int w = selection[t0 + j];
            if (tid == 0 && st.leader[st.cls[w]] == w) st.leader[st.cls[w]] = st.next_twin[w];
            remove_from_rows(w, t0 + j, n, drop, sorted_val, sorted_idx, rank_t, dist_t, st);
            __syncthreads();
        }
        if (tid == 0) st.words[0] = t0 + k + 1;
        return;
    }
    if (tid == 0) st.words[0] = t0 + picks;
}

}  // namespace

template <typename K>
static int segment_sort(K* keys, int64_t n_segments, int64_t n_pad, hipStream_t stream) {
    BYZ_REQUIRE(keys && n_segments > 0 && n_pad >= 2 && (n_pad & (n_pad - 1)) == 0, "segment sort: bad arguments");
    constexpr int CH = chunk_keys<K>();
    const int64_t chunks = n_pad < CH ? 1 : n_pad / CH;
    BYZ_REQUIRE(n_segments * chunks <= 0x7fffffff && ceil_div(n_segments * (n_pad / 2), 256) <= 0x7fffffff,
                "segment sort: too many keys for one launch");
    const unsigned local_grid = static_cast<unsigned>(n_segments * chunks);
    segment_sort_local_kernel<K, true><<<local_grid, kSortThreads, 0, stream>>>(keys, n_pad, static_cast<int>(chunks), 0);
    BYZ_TRY(check_launch("segment_sort_local_kernel"));
    const int64_t n_pairs = n_segments * (n_pad / 2);
    for (int64_t k = 2 * static_cast<int64_t>(CH); k <= n_pad; k <<= 1) {
        for (int64_t j = k / 2; j >= CH; j >>= 1) {
            segment_sort_global_kernel<K><<<static_cast<unsigned>(ceil_div(n_pairs, 256)), 256, 0, stream>>>(keys, n_pad, n_pairs, k, j);
            BYZ_TRY(check_launch("segment_sort_global_kernel"));
        }
        segment_sort_local_kernel<K, false><<<local_grid, kSortThreads, 0, stream>>>(keys, n_pad, static_cast<int>(chunks), k);
        BYZ_TRY(check_launch("segment_sort_local_kernel"));
    }
    return BYZ_OK;
}

int segment_sort_u64(byz_ctx* ctx, unsigned long long* keys, int64_t n_segments, int64_t n_pad, hipStream_t stream) {
    (void)ctx;
    return segment_sort<unsigned long long>(keys, n_segments, n_pad, stream);
}
int segment_sort_u32(byz_ctx* ctx, uint32_t* keys, int64_t n_segments, int64_t n_pad, hipStream_t stream) {
    (void)ctx;
    return segment_sort<uint32_t>(keys, n_segments, n_pad, stream);
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
    if (!want_tables) {      // Krum alone: 32-bit keys, twice the rows per batch
        uint32_t* keys32 = ctx->large_keys.as<uint32_t>();
        int64_t rows32 = 2 * rows_per_batch < n ? 2 * rows_per_batch : n;
        if (rows32 > 32768) rows32 = 32768;
        for (int64_t row0 = 0; row0 < n; row0 += rows32) {
            const int64_t rows = n - row0 < rows32 ? n - row0 : rows32;
            large_row_keys32_kernel<<<dim3(static_cast<unsigned>(ceil_div(n_pad, 256)), static_cast<unsigned>(rows)), 256, 0, stream>>>(
                dist, (int)n, n_pad, (int)row0, keys32);
            BYZ_TRY(check_launch("large_row_keys32_kernel"));
            BYZ_TRY(segment_sort_u32(ctx, keys32, rows, n_pad, stream));
            large_row_scores32_kernel<<<static_cast<unsigned>(rows), 64, 0, stream>>>(keys32, n_pad, (int)row0, (int)prefix_len,
                                                                                      ctx->scores.as<float>());
            BYZ_TRY(check_launch("large_row_scores32_kernel"));
        }
        return BYZ_OK;
    }
    for (int64_t row0 = 0; row0 < n; row0 += rows_per_batch) {
        const int64_t rows = n - row0 < rows_per_batch ? n - row0 : rows_per_batch;
        large_row_keys_kernel<<<dim3(static_cast<unsigned>(ceil_div(n_pad, 256)), static_cast<unsigned>(rows)), 256, 0, stream>>>(
            dist, (int)n, n_pad, (int)row0, keys);
        BYZ_TRY(check_launch("large_row_keys_kernel"));
        BYZ_TRY(segment_sort_u64(ctx, keys, rows, n_pad, stream));
        large_row_tables_kernel<<<static_cast<unsigned>(rows), 256, 0, stream>>>(
            keys, (int)n, n_pad, (int)row0, (int)prefix_len, (int)drop_count, ctx->scores.as<float>(), ctx->sorted_val.as<float>(),
            ctx->large_idx.as<uint32_t>(), ctx->large_rank.as<uint32_t>(), st);
        BYZ_TRY(check_launch("large_row_tables_kernel"));
    }
    {
        // the winner's rank in every row and its distance to every row, as ROWS: what a pick reads (remove_from_rows)
        BYZ_TRY(ctx->large_rank_t.ensure(static_cast<size_t>(n) * n * sizeof(uint32_t)));
        BYZ_TRY(ctx->large_dist_t.ensure(static_cast<size_t>(n) * n * sizeof(float)));
        const unsigned tiles = static_cast<unsigned>(ceil_div(n, 64));
        transpose32_kernel<<<dim3(tiles, tiles), 256, 0, stream>>>(ctx->large_rank.as<uint32_t>(), (int)n, ctx->large_rank_t.as<uint32_t>());
        transpose32_kernel<<<dim3(tiles, tiles), 256, 0, stream>>>(reinterpret_cast<const uint32_t*>(dist), (int)n,
                                                                   ctx->large_dist_t.as<uint32_t>());
        BYZ_TRY(check_launch("transpose32_kernel"));
    }
    return BYZ_OK;
}

int launch_bulyan_loop_large(byz_ctx* ctx, const float* dist, int64_t n, int64_t theta, int64_t drop_count, int64_t users_count,
                             int64_t corrupted, const int32_t* twin_class, int32_t* selection_dev, int32_t* status_dev,
                             hipStream_t stream) {
    BYZ_REQUIRE(dist && selection_dev && status_dev && n > 0 && theta >= 0 && theta <= n,
                "bulyan loop (large): bad arguments (n=%lld theta=%lld)", (long long)n, (long long)theta);
    BYZ_REQUIRE(ctx->large_state.bytes >= large_state_bytes(n) && ctx->large_idx.ptr && ctx->large_rank_t.ptr && ctx->large_dist_t.ptr,
                "bulyan loop (large): the row sort has not run");
    LargeState st = large_state(ctx->large_state.ptr, n);
    st.cls = twin_class;
    BYZ_REQUIRE(twin_class != nullptr, "bulyan loop (large): no twin classes");
    // BYZ_LARGE_BATCH=<k>: picks decided on the exact scores before their contenders are scored together (default 16, at most 32;
    // 1: every pick settled before the next one is decided).  The same selection, pick for pick.
    int batch = 16;
    if (const char* e = std::getenv("BYZ_LARGE_BATCH")) batch = std::atoi(e);
    batch = batch < 1 ? 1 : (batch > kBatchMax ? kBatchMax : batch);
    BYZ_HIP(hipMemsetAsync(status_dev, 0, 3 * sizeof(int32_t), stream));   // status, rows re-scored, (unused)
    BYZ_HIP(hipMemsetAsync(st.words, 0, 4 * sizeof(int32_t), stream));
    // The deciding kernel on many workgroups (a cooperative launch) where the device offers it; BYZ_LARGE_COOP=0: on one workgroup
    // (the comparison, and the form for a device that cannot hold the grid at once).  The same selection.
    int coop = 0;
    BYZ_HIP(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, ctx->device));
    if (const char* e = std::getenv("BYZ_LARGE_COOP")) coop = coop != 0 && std::atoi(e) != 0 ? 1 : 0;
    int grid_blocks = static_cast<int>(ceil_div(n, kGridBlock));
    if (grid_blocks > kGridMaxBlocks) grid_blocks = kGridMaxBlocks;
    if (grid_blocks > ctx->num_cus) grid_blocks = ctx->num_cus;
    const size_t list_bytes = static_cast<size_t>(ceil_div(n, static_cast<int64_t>(grid_blocks) * kGridBlock)) * kGridBlock * sizeof(int32_t);
    if (list_bytes > 60 * 1024) coop = 0;
    BYZ_TRY(ctx->large_grid.ensure(static_cast<size_t>(kGridMaxBlocks) * sizeof(Guess) + 64));
    Guess* wg_best = ctx->large_grid.as<Guess>();
    int32_t* n_listed_dev = reinterpret_cast<int32_t*>(wg_best + kGridMaxBlocks);
    large_scale_kernel<<<1, 1024, 0, stream>>>((int)n, st);      // (timed by the caller: launch_bulyan_loop's KernelTimer)
    BYZ_TRY(check_launch("large_scale_kernel"));
    BYZ_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(st.leader_kept), 0x7fffffff, static_cast<size_t>(n), stream));
    BYZ_HIP(hipMemsetAsync(st.first_kept, 0, static_cast<size_t>(n) * sizeof(int32_t), stream));
    large_twin_leader_kernel<<<static_cast<unsigned>(ceil_div(n, 256)), 256, 0, stream>>>((int)n, st);
    large_twin_links_kernel<<<static_cast<unsigned>(ceil_div(n, 256)), 256, 0, stream>>>((int)n, st);
    BYZ_TRY(check_launch("large_twin_links_kernel"));
    const unsigned rescore_grid = static_cast<unsigned>(ctx->num_cus) * 4;      // 16 waves per CU
    // The host does not know where a batch was cut: it queues as many batches as the picks left would take if every batch stood,
    // a few more, and looks at the next pick; queued batches behind the last pick leave at once.
    BYZ_TRY(ctx->pinned.ensure(4 * sizeof(int32_t)));
    int64_t next = 0;
    for (int round = 0; next < theta; ++round) {
        const int64_t queued = ceil_div(theta - next, batch) + 4;
        for (int64_t q = 0; q < queued; ++q) {
            if (coop != 0) {
                int n_i = (int)n, theta_i = (int)theta, drop_i = (int)drop_count, users_i = (int)users_count, corrupted_i = (int)corrupted;
                const float* val_p = ctx->sorted_val.as<float>();
                const uint32_t* idx_p = ctx->large_idx.as<uint32_t>();
                const uint32_t* rank_p = ctx->large_rank_t.as<uint32_t>();
                const float* dist_p = ctx->large_dist_t.as<float>();
                LargeState st_arg = st;
                const int32_t* status_p = status_dev;
                int32_t* rescored_p = status_dev + 1;
                void* args[] = {&n_i, &theta_i, &drop_i, &users_i, &corrupted_i, &batch, &val_p, &idx_p, &rank_p, &dist_p, &st_arg,
                                &status_p, &rescored_p, &wg_best, &n_listed_dev};
                BYZ_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&large_decide_grid_kernel), dim3(grid_blocks),
                                                   dim3(kGridBlock), args, static_cast<unsigned>(list_bytes), stream));
            } else {
                large_decide_kernel<<<1, 1024, 0, stream>>>((int)n, (int)theta, (int)drop_count, (int)users_count, (int)corrupted, batch,
                                                            ctx->sorted_val.as<float>(), ctx->large_idx.as<uint32_t>(),
                                                            ctx->large_rank_t.as<uint32_t>(), ctx->large_dist_t.as<float>(), st, status_dev,
                                                            status_dev + 1);
            }
            large_rescore_kernel<<<rescore_grid, 256, 0, stream>>>((int)n, (int)users_count, (int)corrupted, ctx->sorted_val.as<float>(),
                                                                   ctx->large_idx.as<uint32_t>(), st, status_dev);
            large_settle_kernel<<<1, 1024, 0, stream>>>((int)n, (int)theta, (int)drop_count, ctx->sorted_val.as<float>(),
                                                        ctx->large_idx.as<uint32_t>(), ctx->large_rank_t.as<uint32_t>(),
                                                        ctx->large_dist_t.as<float>(), st, selection_dev, status_dev);
        }
        BYZ_TRY(check_launch("large_decide_kernel / large_rescore_kernel / large_settle_kernel"));
        int32_t* host = static_cast<int32_t*>(ctx->pinned.ptr);
        BYZ_HIP(hipMemcpyAsync(host, st.words, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        BYZ_HIP(hipMemcpyAsync(host + 1, status_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        BYZ_HIP(hipStreamSynchronize(stream));
        if (host[1] != 0) break;                 // the reference's KeyError: the caller reads the status word
        BYZ_REQUIRE(host[0] > next || host[0] >= theta, "bulyan loop (large): no pick settled in %lld batches", (long long)queued);
        next = host[0];
    }
    return BYZ_OK;
}

}  // namespace byz
