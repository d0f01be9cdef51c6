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
// bulyan_grid_kernel: all theta picks in one launch, rows spread over ceil(n / 256) workgroups that exchange
// 8-byte tagged granules once per pick.  Re-sorting every remaining row per pick, as the reference does, costs
// O(theta N^2 log N); here each row keeps two running fp64 sums: T = sum of its distances to the rows still
// present, and Top = sum of the `drop` largest of them (drop = f - 1 when users_count == N).  The exact value of
// the reference's score, "sum of the n_t - f smallest of the n_t - 1 remaining distances", is T - Top, and
// removing the winner w updates both in O(1) per row through the precomputed rank table.  The reference itself
// forms that sum sequentially in fp32; whenever more than one twin class lies within the rounding band of such a
// sum, the contenders are re-scored in exactly that arithmetic (reference_score), so that given the same distance
// matrix the selection is the reference's, pick for pick -- not a more accurate one.
#include "common.hpp"

#include <cstdlib>
#include <cstring>

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

// (defined with the Bulyan re-score below; row_sort_kernel's Krum score uses it too)
__device__ __forceinline__ float integer_passes(uint32_t (&M)[8], const int (&ex)[8], float s, int lane, unsigned long long& n_passes);

// ---- the row sort's bitonic network, register-blocked (round 6) ----------------------------------------------------------
// The textbook form below (one compare-exchange level per pass over LDS, a barrier behind each) takes log2(n)(log2(n) + 1) / 2
// passes: 105 at n_pad = 16,384, every one a full round trip of the 128 KB of keys through the LDS pipe -- 200 us per row, 7.9 ms
// for the 10,000 rows of configs[4].  Here a work item takes 16 keys into registers and runs up to FOUR levels of the network
// on them before they go back: for the merge of size k = 2^m the levels j = 2^(m-1) .. 1 are cut, from the bottom, into groups
// of four (j = 8 .. 1 on 16 contiguous keys, j = 128 .. 16 on keys 16 apart, j = 2048 .. 256, ...; the top group takes the
// levels that are left), and the merges k = 2 .. 16 are one pass: 32 passes instead of 105, the same compare-exchanges, the same
// network -- the sorted order of the 64-bit keys is unique, so the result is the textbook form's bit for bit.  The direction of
// a compare-exchange, (index & k) == 0, is uniform per work item from k = 16 on (k lies above every bit in which the item's
// keys differ) and a compile-time function of the slot for k = 2, 4, 8 inside the first pass.
// Layout: key i sits at word i + (i >> 4) (one word of padding per 16): a work item's 16 CONTIGUOUS keys then start 17 words
// apart, so that the 64 lanes of a wave spread over all banks; with keys 16 or more apart the lanes of a wave read consecutive
// words anyway.
__device__ __forceinline__ int sort_slot(int i) { return i + (i >> 4); }

// Inside the network a key is a DOUBLE: (order-preserving float bits << 16 | column) is an integer below 2^48, exact in a
// double, and doubles of one sign order like their integers -- so a compare-exchange is v_min_f64 + v_max_f64 (full rate on this
// part) instead of a 64-bit compare and four selects, which is what bounded the network on 64-bit integer keys (five vector
// instructions per compare-exchange, 860,000 of them per row at 16,384 keys).  A work item that sorts DESCENDING flips the sign
// of its keys on the way in and out (one xor per key) and sorts ascending in between.
__device__ __forceinline__ double key_to_double(uint32_t ordered, uint32_t column) {
    return static_cast<double>((static_cast<unsigned long long>(ordered) << 16) | column);
}
__device__ __forceinline__ unsigned long long double_to_key(double d) {     // -> ordered bits << 32 | column
    const unsigned long long x = static_cast<unsigned long long>(d);
    return ((x >> 16) << 32) | (x & 0xffffull);
}
__device__ __forceinline__ void compare_exchange_up(double& a, double& b) {
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo;
    b = hi;
}
__device__ __forceinline__ double flip_sign(double v, bool flip) {
    return __longlong_as_double(__double_as_longlong(v) ^ (flip ? static_cast<long long>(0x8000000000000000ull) : 0ll));
}

// LV levels (strides 2^(LV-1) .. 1 in units of `stride` keys) of the merge of size k on the 2^LV keys base + s * stride
template <int LV>
__device__ __forceinline__ void sort_levels(double* keys, int base, int stride, int k) {
    constexpr int E = 1 << LV;
    double r[E];
    const bool down = (base & k) != 0;       // k lies above every bit in which the item's keys differ
#pragma unroll
    for (int s = 0; s < E; ++s) r[s] = flip_sign(keys[sort_slot(base + s * stride)], down);
#pragma unroll
    for (int l = LV - 1; l >= 0; --l) {
#pragma unroll
        for (int s = 0; s < E; ++s) {
            if ((s & (1 << l)) == 0) compare_exchange_up(r[s], r[s | (1 << l)]);
        }
    }
#pragma unroll
    for (int s = 0; s < E; ++s) keys[sort_slot(base + s * stride)] = flip_sign(r[s], down);
}

// n_pad >= 256 keys (a power of two) as doubles at keys[sort_slot(i)]: sorted ascending and left as 64-bit integer keys
// (order-preserving float bits << 32 | column); every thread of the workgroup calls it
__device__ __forceinline__ void blocked_bitonic_sort(unsigned long long* keys_u64, int n_pad, int tid, int nt) {
    double* const keys = reinterpret_cast<double*>(keys_u64);
    const int items = n_pad >> 4;
    // merges k = 2 .. 16: 16 contiguous keys per item, every level in registers
    for (int w = tid; w < items; w += nt) {
        const int base = w << 4;
        double r[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = keys[sort_slot(base + s)];
#pragma unroll
        for (int kb = 1; kb <= 3; ++kb) {          // k = 2, 4, 8: the direction is a compile-time function of the slot
#pragma unroll
            for (int l = kb - 1; l >= 0; --l) {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    if ((s & (1 << l)) == 0) {
                        if ((s & (1 << kb)) == 0) compare_exchange_up(r[s], r[s | (1 << l)]);
                        else compare_exchange_up(r[s | (1 << l)], r[s]);
                    }
                }
            }
        }
        const bool down = (base & 16) != 0;        // k = 16
#pragma unroll
        for (int s = 0; s < 16; ++s) r[s] = flip_sign(r[s], down);
#pragma unroll
        for (int l = 3; l >= 0; --l) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if ((s & (1 << l)) == 0) compare_exchange_up(r[s], r[s | (1 << l)]);
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) keys[sort_slot(base + s)] = flip_sign(r[s], down);
    }
    __syncthreads();
    for (int m = 5; (1 << m) <= n_pad; ++m) {       // the merge of size k = 2^m: levels j = 2^(m-1) .. 1
        const int k = 1 << m;
        int top = m;                                // levels still to do: j = 2^(top-1) .. 1
        // the top group takes what is left above the groups of four
        const int first = (m & 3) == 0 ? 4 : (m & 3);
        {
            const int lo = top - first;             // this group's smallest stride is 2^lo
            const int stride = 1 << lo;
            const int group_items = n_pad >> first;
            for (int w = tid; w < group_items; w += nt) {
                const int base = ((w >> lo) << (lo + first)) | (w & (stride - 1));
                if (first == 4) sort_levels<4>(keys, base, stride, k);
                else if (first == 3) sort_levels<3>(keys, base, stride, k);
                else if (first == 2) sort_levels<2>(keys, base, stride, k);
                else sort_levels<1>(keys, base, stride, k);
            }
            __syncthreads();
            top = lo;
        }
        while (top > 0) {                           // groups of four levels: strides 2^(top-1) .. 2^(top-4)
            const int lo = top - 4;
            const int stride = 1 << lo;
            for (int w = tid; w < items; w += nt) {
                const int base = ((w >> lo) << (lo + 4)) | (w & (stride - 1));
                sort_levels<4>(keys, base, stride, k);
            }
            __syncthreads();
            top = lo;
        }
    }
    // back to integer keys for everything behind the sort
    for (int w = tid; w < items; w += nt) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int at = sort_slot((w << 4) + s);
            keys_u64[at] = double_to_key(keys[at]);
        }
    }
    __syncthreads();
}

// BLOCKED: the register-blocked network on the padded layout (n_pad >= 256); otherwise the textbook form
template <bool TABLES, bool BLOCKED>
__global__ void row_sort_kernel(const float* __restrict__ dist, int n, int n_pad, int prefix_len, int drop,
                                float* __restrict__ scores, uint16_t* __restrict__ sorted_idx,
                                uint16_t* __restrict__ rank_rows, double* __restrict__ row_total,
                                double* __restrict__ row_top, float* __restrict__ sorted_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys_raw[];  // n_pad keys (+ padding), then scratch
    const int key_words = BLOCKED ? n_pad + (n_pad >> 4) : n_pad;
    double* scratch = reinterpret_cast<double*>(keys_raw + key_words);           // blockDim.x doubles
    // keys[i]: the i-th key, wherever the layout puts it
    struct KeyArray {
        unsigned long long* p;
        __device__ __forceinline__ unsigned long long& operator[](int i) const { return p[BLOCKED ? sort_slot(i) : i]; }
    };
    const KeyArray keys{keys_raw};
    const int u = blockIdx.x;
    const int tid = threadIdx.x;
    const int nt = blockDim.x;

    const float* row = dist + static_cast<int64_t>(u) * n;
    for (int c = tid; c < n_pad; c += nt) {
        unsigned long long key;
        if (c >= n) {
            key = ~0ull;
        } else {
            // the self entry sorts behind every real one, +inf and NaN included (a client with a non-finite gradient)
            // NaN of either sign: behind +inf (the Gram identity makes inf - inf of an infinite gradient, whose sign is anybody's)
            const float d = row[c];
            const uint32_t ob = (c == u) ? 0xffffffffu : (d != d ? 0xfffffffeu : ordered_bits(d));
            key = (static_cast<unsigned long long>(ob) << 32) | static_cast<unsigned>(c);
        }
        if constexpr (BLOCKED) {
            // (the network sorts doubles: see key_to_double; a padding key's column is 0xffff, above every real column)
            const double as_double = key_to_double(static_cast<uint32_t>(key >> 32), c >= n ? 0xffffu : static_cast<uint32_t>(c));
            keys_raw[sort_slot(c)] = static_cast<unsigned long long>(__double_as_longlong(as_double));
        } else {
            keys[c] = key;
        }
    }
    __syncthreads();

    if constexpr (BLOCKED) {
        blocked_bitonic_sort(keys_raw, n_pad, tid, nt);
    } else {
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
    }
    // The real neighbours occupy ranks 0 .. n-2 (a NaN with an all-ones payload could tie with the self entry's key; the
    // sums below skip the self entry by its column, not by its rank).

    // The Krum score: the sequential fp32 sum of the first prefix_len sorted values, exactly as Python's sum() forms it
    // (defences.py:33-34).  One thread walking the prefix out of LDS took ~70 cycles per entry -- at N = 10,000 more than the
    // sort itself; wave 0 now adds the first 512 entries as a chain from broadcast reads and the rest in integer passes
    // (integer_passes below: the same bits).  A prefix that holds a sign bit, or a sum that leaves the finite range, is summed
    // again the old way.
    // (an empty prefix -- users_count == corrupted_count, reachable through return_index=True which skips the assert --
    // stores sum([]) == 0.0 like the reference: the guard must not skip the store)
    if (tid < 64) {
        const int lane = tid;
        float s = 0.0f;
        const int head_n = prefix_len < 512 ? (prefix_len > 0 ? prefix_len : 0) : 512;
        for (int r0 = 0; r0 < head_n; r0 += 16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = r0 + i;
                v[i] = r < head_n ? from_ordered_bits(static_cast<uint32_t>(keys[r] >> 32)) : 0.0f;   // (+ 0.0 is exact)
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) s = __fadd_rn(s, v[i]);
        }
        bool odd = false;   // a negative value, or -0.0: the passes take non-negative distances
        for (int r = lane; r < head_n; r += 64) odd = odd || (from_ordered_bits(static_cast<uint32_t>(keys[r] >> 32)) < 0.0f);
        unsigned long long n_passes = 0;
        for (int r0 = 512; r0 < prefix_len && __ballot(odd) == 0ull; r0 += 512) {
            uint32_t M[8];
            int ex[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = r0 + 8 * lane + j;
                const uint32_t xb = r < prefix_len ? __float_as_uint(from_ordered_bits(static_cast<uint32_t>(keys[r < n_pad ? r : n_pad - 1] >> 32))) : 0u;
                odd = odd || (xb >> 31) != 0u;
                const uint32_t e = (xb >> 23) & 0xffu;
                ex[j] = e != 0u ? static_cast<int>(e) : 1;
                M[j] = e != 0u ? ((xb & 0x7fffffu) | 0x800000u) : (xb & 0x7fffffu);
            }
            if (__ballot(odd) == 0ull) {
                s = integer_passes(M, ex, s, lane, n_passes);
            }
        }
        if (__ballot(odd) != 0ull || !(__builtin_fabsf(s) <= 3.4028234663852886e38f)) {
            s = 0.0f;
            for (int r = 0; r < prefix_len; ++r) s = __fadd_rn(s, from_ordered_bits(static_cast<uint32_t>(keys[r] >> 32)));
        }
        if (tid == 0) scores[u] = s;
    }

    if (TABLES) {
        // tot / top are sums over the FINITE distances; the non-finite ones (+inf, NaN: a client whose gradient is not
        // finite) sort last and are counted -- a row's score is finite exactly while they all lie among its `drop` largest
        // entries, which is what the reference's sorted(...)[:k] sum gives for +inf (defences.py:33-34)
        //
        // The rank of every column in this row goes out as a ROW (rank_rows[u][c], contiguous) and rank_transpose_kernel turns
        // the rows into rank_t[c][u] afterwards.  Round 6: written straight into rank_t, the ranks were 2-byte stores n entries
        // apart -- N^2 of them, every one a partial line that 64 different workgroups touch at different times: 0.38 GB of
        // fabric writes for 0.13 GB of tables at N = 4000 by the PMC, and what the kernel's time was made of (the sort itself
        // was a third of it).  The row is staged in LDS over the key array: a thread first takes its keys into registers, and
        // only when every thread has (and wave 0 has formed the score from the keys) do the ranks overwrite them.
        constexpr int kMine = 16;                      // n <= 16,384 keys over >= n_pad / 16 threads
        unsigned long long mine[kMine];
#pragma unroll
        for (int i = 0; i < kMine; ++i) {
            const int r = tid + i * nt;
            mine[i] = r < n ? keys[r] : ~0ull;
        }
        double tot = 0.0, top = 0.0, bad = 0.0;
        const int first_top = n - 1 - drop;
#pragma unroll
        for (int i = 0; i < kMine; ++i) {
            const int r = tid + i * nt;
            if (r < n) {
                const unsigned long long key = mine[i];
                const int c = static_cast<int>(key & 0xffffffffu);
                sorted_idx[static_cast<int64_t>(u) * n + r] = static_cast<uint16_t>(c);
                const float v = c == u ? __builtin_inff() : from_ordered_bits(static_cast<uint32_t>(key >> 32));
                sorted_val[static_cast<int64_t>(u) * n + r] = v == 0.0f ? 0.0f : v;   // never -0.0: that bit pattern marks a removed entry
                if (c != u) {
                    if (__builtin_fabsf(v) <= 3.4028234663852886e38f) {
                        tot += static_cast<double>(v);
                        if (r >= first_top) top += static_cast<double>(v);
                    } else {
                        bad += 1.0;
                    }
                }
            }
        }
        __syncthreads();     // every key is in a register and the score is formed: the key array is free
        uint16_t* const rank_row = reinterpret_cast<uint16_t*>(keys_raw);
#pragma unroll
        for (int i = 0; i < kMine; ++i) {
            const int r = tid + i * nt;
            if (r < n) rank_row[static_cast<int>(mine[i] & 0xffffffffu)] = static_cast<uint16_t>(r);
        }
        __syncthreads();
        for (int c = tid; c < n; c += nt) rank_rows[static_cast<int64_t>(u) * n + c] = rank_row[c];
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
        __syncthreads();
        scratch[tid] = bad;
        __syncthreads();
        for (int s = nt >> 1; s > 0; s >>= 1) {
            if (tid < s) scratch[tid] += scratch[tid + s];
            __syncthreads();
        }
        if (tid == 0) row_top[n + u] = scratch[0];   // the count of non-finite entries rides behind the n sums
    }
}

// rank_t[c][u] = rank_rows[u][c]: 64 x 64 tiles through LDS, both sides in contiguous runs
__global__ __launch_bounds__(256) void rank_transpose_kernel(const uint16_t* __restrict__ rank_rows, int n, uint16_t* __restrict__ rank_t) {
    __shared__ uint16_t tile[64][66];
    const int u0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int uu = u0 + ty + 4 * i, cc = c0 + tx;
        if (uu < n && cc < n) tile[ty + 4 * i][tx] = rank_rows[static_cast<int64_t>(uu) * n + cc];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = c0 + ty + 4 * i, uu = u0 + tx;
        if (uu < n && cc < n) rank_t[static_cast<int64_t>(cc) * n + uu] = tile[tx][ty + 4 * i];
    }
}

// ---------------------------------------------------------------------------------------------------
struct Candidate {
    double score;
    int pos;   // position in the reference's visit order; INT_MAX = none
    int row;
    int cls;   // the row's twin class where the caller needs it (the Bulyan loop), else 0: travels with the winner
};

__device__ __forceinline__ bool better(const Candidate& a, const Candidate& b) {
    // strict '<' on the score; an equal score keeps the earlier visitor
    return a.score < b.score || (a.score == b.score && a.pos < b.pos);
}

// Wave-wide reductions without the LDS pipe: four DPP steps inside every 16-lane row (xor 1, xor 2, half mirror, mirror: each
// an involution, so every lane ends up with its row's result), then the four rows' results through v_readlane.  A
// ds_bpermute shuffle costs ~100 cycles of dependent latency and the loop makes ~30 of them per pick; a DPP move ~8.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>(static_cast<int>(b)), hi = dpp_i<CTRL>(static_cast<int>(b >> 32));
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(b), l), hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), l);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned>(lo));
}

__device__ __forceinline__ Candidate wave_best(Candidate c) {   // wave-uniform result
#define BYZ_STEP(CTRL)                                                                              \
    {                                                                                               \
        Candidate o;                                                                                \
        o.score = dpp_d<CTRL>(c.score);                                                             \
        o.pos = dpp_i<CTRL>(c.pos);                                                                 \
        o.row = dpp_i<CTRL>(c.row);                                                                 \
        o.cls = dpp_i<CTRL>(c.cls);                                                                 \
        if (better(o, c)) c = o;                                                                    \
    }
    BYZ_STEP(0xB1) BYZ_STEP(0x4E) BYZ_STEP(0x141) BYZ_STEP(0x140)
#undef BYZ_STEP
    Candidate best{readlane_d(c.score, 0), __builtin_amdgcn_readlane(c.pos, 0), __builtin_amdgcn_readlane(c.row, 0),
                   __builtin_amdgcn_readlane(c.cls, 0)};
#pragma unroll
    for (int l = 16; l < 64; l += 16) {
        const Candidate o{readlane_d(c.score, l), __builtin_amdgcn_readlane(c.pos, l), __builtin_amdgcn_readlane(c.row, l),
                          __builtin_amdgcn_readlane(c.cls, l)};
        if (better(o, best)) best = o;
    }
    return best;
}
__device__ __forceinline__ double wave_min_d(double v) {   // wave-uniform; NaN-free inputs
    v = fmin(v, dpp_d<0xB1>(v));
    v = fmin(v, dpp_d<0x4E>(v));
    v = fmin(v, dpp_d<0x141>(v));
    v = fmin(v, dpp_d<0x140>(v));
    return fmin(fmin(readlane_d(v, 0), readlane_d(v, 16)), fmin(readlane_d(v, 32), readlane_d(v, 48)));
}
__device__ __forceinline__ float wave_min_f(float v) {
    v = __builtin_fminf(v, __int_as_float(dpp_i<0xB1>(__float_as_int(v))));
    v = __builtin_fminf(v, __int_as_float(dpp_i<0x4E>(__float_as_int(v))));
    v = __builtin_fminf(v, __int_as_float(dpp_i<0x141>(__float_as_int(v))));
    v = __builtin_fminf(v, __int_as_float(dpp_i<0x140>(__float_as_int(v))));
    const int b = __float_as_int(v);
    return __builtin_fminf(__builtin_fminf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                           __builtin_fminf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<0xB1>(v));
    v = min(v, dpp_i<0x4E>(v));
    v = min(v, dpp_i<0x141>(v));
    v = min(v, dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
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
// Twin classes.  Two rows u, v are twins when d(u, v) == 0 and d(u, x) == d(v, x) bitwise for every other x:
// what two clients that submitted the same vector look like (malicious.py:26-27 rebinds every malicious
// client's gradient to ONE array).  Twins keep identical live distance multisets through every removal, so in
// the reference their scores are identical floats at every pick and only the visit order separates them.
// cls[u] = the smallest row of u's class.  One wave per row: the first zero in the row nominates, a full
// bitwise comparison of the two rows decides (an arbitrary caller-supplied matrix need not be a metric).
__global__ __launch_bounds__(256) void twin_class_kernel(const float* __restrict__ dist, int n,
                                                         int32_t* __restrict__ cls) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (u >= n) return;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(dist) + static_cast<int64_t>(u) * n;
    int cand = -1;
    for (int j0 = 0; j0 < u; j0 += 64) {
        const int j = j0 + lane;
        const unsigned long long m = __ballot(j < u && (row[j] << 1) == 0u);   // +0.0 or -0.0
        if (m) {
            cand = j0 + __builtin_ctzll(m);
            break;
        }
    }
    int result = u;
    if (cand >= 0) {
        const uint32_t* other = reinterpret_cast<const uint32_t*>(dist) + static_cast<int64_t>(cand) * n;
        bool same = true;
        for (int x = lane; x < n; x += 64)
            if (x != u && x != cand && row[x] != other[x]) same = false;
        if ((other[u] << 1) != 0u) same = false;
        if (__ballot(!same) == 0ull) result = cand;
    }
    if (lane == 0) cls[u] = result;
}

// a class root must be its own root (always true for genuine twins; an inconsistent matrix falls back to singletons)
__global__ __launch_bounds__(256) void twin_class_fix_kernel(const int32_t* __restrict__ cls, int n,
                                                             int32_t* __restrict__ out) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n) return;
    const int r = cls[u];
    out[u] = (cls[r] == r) ? r : u;
}

// ---------------------------------------------------------------------------------------------------
// Bulyan's pick-and-remove loop (defences.py:59-68) across workgroups.
//
// Workgroup g of W = ceil(n / 256) owns rows 256 g .. 256 g + 255, one row per thread.  Every row keeps two fp64
// running sums: T (its distances to the rows still present) and Top (the `drop` largest of them); T - Top is
// the EXACT value of the sum the reference forms in fp32 at that pick.  Each pick:
//
//   1. every workgroup finds its best row (smallest T - Top, then earliest in the visit order 1, 0, 2, ...) and
//      the best score among its rows of any OTHER twin class, and publishes both as 8-byte tagged granules
//      (one relaxed agent-scope store each: the payload carries its own tag, so no fence and no flag);
//   2. wave 0 of every workgroup gathers all W granules.  The reference decides by SEQUENTIAL fp32 sums, whose
//      rounding error is at most delta = u (m + 1) / 2 relative for m ascending positive terms (u = 2^-24):
//      a row whose exact score exceeds the minimum by more than ~2.2 delta cannot win in the reference either.
//      If every row within that band belongs to one twin class, the winner is that class's earliest member in
//      the visit order -- no fp32 arithmetic needed (this is every pick of well-separated data, and every pick
//      among the attack's identical rows);
//   3. otherwise (round 2) the contenders ARE re-scored the reference's way -- ascending live distances, a
//      left-to-right fp32 sum of the first n_t - f (defences.py:33-34) -- one wave per contender, one
//      contender per twin class and workgroup, in parallel across the workgroups; the fp32 scores are
//      gathered the same way and the smallest, earliest one wins: bit for bit the reference's decision.
//   4. everybody removes the winner: O(1) per row through the rank table.
//
// All workgroups must be resident (W <= 64 of 256 CUs); every wait is bounded and a timeout is reported through
// the status word, never papered over.
constexpr int kGridThreads = 256;
constexpr int kGridMaxWgs = kMaxSelectRows / kGridThreads;   // 64: one lane of the gathering wave per workgroup
constexpr unsigned kSpinLimit = 1u << 22;
constexpr uint32_t kNoRow = 0x3fffu;
constexpr uint32_t kInfBits = 0x7f800000u;

__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// largest float <= d (d >= 0 or +inf; NaN -> +inf, "no candidate")
__device__ __forceinline__ float float_below(double d) {
    if (!(d == d)) return __builtin_inff();
    float f = static_cast<float>(d);
    if (static_cast<double>(f) > d) {
        const uint32_t b = __float_as_uint(f);
        f = __uint_as_float((b & 0x7fffffffu) == 0u ? 0x80000001u : ((b & 0x80000000u) ? b + 1u : b - 1u));
    }
    return f;
}
__device__ __forceinline__ double double_above(float f) {   // an upper bound of every double that rounds down to f
    const uint32_t b = __float_as_uint(f);
    if ((b & 0x7f800000u) == 0x7f800000u) return static_cast<double>(f);
    const float up = (b & 0x80000000u) ? ((b & 0x7fffffffu) == 0u ? __uint_as_float(1u) : __uint_as_float(b - 1u))
                                       : __uint_as_float(b + 1u);
    return static_cast<double>(up);
}

// wave 0 only: lane l < n_wgs waits for workgroup l's granule of this pick; false on timeout
__device__ __forceinline__ bool gather_granules(const unsigned long long* slots, int n_wgs, uint32_t tag_mask,
                                                uint32_t tag, int lane, unsigned long long none,
                                                unsigned long long& mine) {
    unsigned long long v = none;
    bool ok = lane >= n_wgs;
    for (unsigned spins = 0;; ++spins) {
        if (!ok) {
            v = granule_load(slots + lane);
            ok = (static_cast<uint32_t>(v) & tag_mask) == tag;
        }
        if (__ballot(!ok) == 0ull) break;
        if (spins > kSpinLimit) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    mine = v;
    return true;
}

__device__ __forceinline__ int wave_max_int(int v) {   // wave-uniform result
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));   // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));   // row_mirror
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ int wave_sum_int(int x) {   // wave-uniform result
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);   // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
           __builtin_amdgcn_readlane(x, 48);
}

// both granules of a pick in one polling loop: one exchange latency instead of two
__device__ __forceinline__ bool gather_granule_pair(const unsigned long long* slots_a, const unsigned long long* slots_b,
                                                    int n_wgs, uint32_t tag4, uint32_t tag18, int lane,
                                                    unsigned long long& mine_a, unsigned long long& mine_b) {
    unsigned long long va = mine_a, vb = mine_b;
    bool ok_a = lane >= n_wgs, ok_b = lane >= n_wgs;
    for (unsigned spins = 0;; ++spins) {
        if (!ok_a) {
            va = granule_load(slots_a + lane);
            ok_a = (static_cast<uint32_t>(va) & 15u) == tag4;
        }
        if (!ok_b) {
            vb = granule_load(slots_b + lane);
            ok_b = (static_cast<uint32_t>(vb) & 0x3ffffu) == tag18;
        }
        if (__ballot(!(ok_a && ok_b)) == 0ull) break;
        if (spins > kSpinLimit) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    mine_a = va;
    mine_b = vb;
    return true;
}

// The reference's score of row u at this pick (defences.py:33-34): ascending live distances, sequential fp32 sum of the
// first `take`, as a PLAIN dependent chain of fp32 additions.  One wave; the result is wave-uniform.  The row's ascending
// values and their columns come from the tables row_sort_kernel wrote; the entries of FOUR batches of 512 are in flight
// (unconditional loads at clamped positions; what lies past the row is masked when it is used), the entries of a batch
// (zeros for removed columns and past the prefix: adding +0.0 is exact) go to LDS in order, and every lane of the wave adds
// them up left to right from broadcast reads -- the reference's loop, literally.
//
// Since round 3 this is the COMPARISON (BYZ_BULYAN_RESCORE=plain) and the fallback for rows the integer passes refuse (a
// sign bit on a live entry); the default re-score is reference_score_marked below.  Two earlier attempts to beat this form
// were measured, selections identical in every case (profiles/r02q .. r02t, r03a, r03k): round 2's integer rule per 64
// entries with every chunk that held a tie or a binade crossing falling back to a 63-step DPP chain, and a version per 512
// entries with a checkpoint behind every row's first 512 entries: 260 / 234 ms at N = 10,000 against this form's 253.  What
// they lacked is what the passes below have: ties and crossings handled INSIDE the parallel pass, and no staging at all.
__device__ __forceinline__ float reference_score_plain(const float* __restrict__ sorted_val, const uint16_t* __restrict__ sorted_idx,
                                                       const uint32_t* removed, int n, int u, int take, int lane,
                                                       float* __restrict__ stage) {
    constexpr int kDepth = 8;
    constexpr int kBatch = 64 * kDepth;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const uint16_t* order = sorted_idx + static_cast<int64_t>(u) * n;
    const float* vals = sorted_val + static_cast<int64_t>(u) * n;
    float carry = 0.0f;
    int got = 0;
    struct Buf {
        int col[kDepth];
        float v[kDepth];
    };
    auto fetch = [&](Buf& b, int r0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            int r = r0 + 64 * k + lane;
            r = r < n ? r : n - 1;                 // nothing conditional about the load itself
            b.col[k] = order[r];
            b.v[k] = vals[r];
        }
    };
    auto consume = [&](const Buf& b, int r0) __attribute__((always_inline)) {
        int chunks = 0;             // chunks of this batch that hold an entry of the prefix
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            const bool inside = r0 + 64 * k + lane < n;
            const int c = b.col[k];
            const bool live = inside && c != u && !((removed[c >> 5] >> (c & 31)) & 1u);
            const unsigned long long m = __ballot(live);
            const int before = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                         __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
            stage[64 * k + lane] = (live && got + before < take) ? b.v[k] : 0.0f;   // past the prefix: + 0.0 (exact)
            if (m != 0ull && got < take) chunks = k + 1;
            got += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const f32x4* src = reinterpret_cast<const f32x4*>(stage);
        // one chunk per trip: 16 broadcast reads (every lane the same addresses) issued together, then the 64 additions --
        // only the first read's latency is exposed per trip (hipcc folds a hand-pipelined loop back into this shape)
        for (int k = 0; k < chunks; ++k) {
            f32x4 e[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) e[i] = src[16 * k + i];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                carry = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(carry, e[i].x), e[i].y), e[i].z), e[i].w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the batch is read before the next one overwrites it
        __builtin_amdgcn_wave_barrier();
    };
    Buf a, b, c, d;
    fetch(a, 0);
    fetch(b, kBatch);
    fetch(c, 2 * kBatch);
    fetch(d, 3 * kBatch);
    for (int r0 = 0;; r0 += 4 * kBatch) {
        consume(a, r0);
        if (r0 + kBatch >= n || got >= take) break;
        fetch(a, r0 + 4 * kBatch);
        consume(b, r0 + kBatch);
        if (r0 + 2 * kBatch >= n || got >= take) break;
        fetch(b, r0 + 5 * kBatch);
        consume(c, r0 + 2 * kBatch);
        if (r0 + 3 * kBatch >= n || got >= take) break;
        fetch(c, r0 + 6 * kBatch);
        consume(d, r0 + 3 * kBatch);
        if (r0 + 4 * kBatch >= n || got >= take) break;
        fetch(d, r0 + 7 * kBatch);
    }
    return carry;
}

// ---------------------------------------------------------------------------------------------------
// The same score, bit for bit, WITHOUT the chain of dependent additions (round 3; prototype and its checks:
// scripts/proto/seqsum_int.py).  A dependent v_add_f32 costs ~10 cycles on a SIMD that runs one wave, so the literal chain
// over a 7600-entry prefix is ~76,000 cycles however its operands are fed (profiles/r03s_bulyan_pair_rescore.txt).  But
// while the running sum s = I q (q its ulp, 2^23 <= I < 2^24) stays inside its binade, round-to-nearest-even is integer
// arithmetic:
//       fl(s + x) = (I + a + t) q,   a = floor(x / q),   t = [rem > q / 2]  or  [rem == q / 2 and I + a odd],
// and the only thing one entry needs from its predecessors is the PARITY of I in front of it (ties) -- a prefix over
// "xor b" (no tie: b = a + t mod 2) and "reset to 0" (a tie always leaves an even I).  So a wave takes 512 entries at once,
// eight consecutive ones per lane: a and the remainder's class per entry, the lane's increment under either incoming
// parity, the incoming parity of every lane from two ballots, a DPP prefix sum, and the first entry at which I reaches
// 2^24: that ONE entry is added in fp32 (always right, whatever the binades do), and the pass restarts behind it with the
// new q.  A prefix of m entries has ~log2(m / 64) such crossings once the first 64 entries -- where the sum doubles every
// other step -- have been added literally: 23 passes for 7600 entries.
//
// No staging either: the table of ascending values is MARKED as the loop goes -- every row sets the entry of the removed
// winner (its rank comes from the rank table the O(1) update reads anyway) and of its own diagonal to -0.0, which adds
// nothing and is never a live distance (row_sort_kernel stores +0.0 for a zero) -- so a re-score reads the row's entries as
// they lie, counts the live ones for the prefix cut, and needs neither the columns nor the `removed` bitmap.
constexpr uint32_t kGoneBits = 0x80000000u;   // -0.0f

__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v) {   // inclusive, lane order
    int x = static_cast<int>(v);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1 (zeros shifted in)
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return static_cast<uint32_t>(x);
}

// 512 entries onto s (wave-uniform in and out).  Lane l holds entries 8 l .. 8 l + 7 as significand M (0: the entry adds
// nothing -- removed, past the prefix, or already in s) and biased exponent ex >= 1 (subnormals count as exponent 1).
// Per pass and entry, under the unit q of s (biased exponent es): (M : 0) >> (es - ex) as ONE 64-bit shift gives
// a = floor(x / q) in the high word and the remainder, left-aligned, in the low word -- above half a unit iff it exceeds
// 0x80000000, a tie iff it equals it.  Lanes without a tie (nearly all) have their increment at once, and its parity is
// their "xor"; only lanes that hold a tie walk their eight entries under both incoming parities.
__device__ __forceinline__ float integer_passes(uint32_t (&M)[8], const int (&ex)[8], float s, int lane, unsigned long long& n_passes) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (;;) {
        // (s is the same in every lane; saying so keeps the pass's bookkeeping on the scalar unit and its branches uniform)
        const uint32_t sbits = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(__float_as_uint(s))));
        const int e0 = static_cast<int>(sbits >> 23);
        if (e0 >= 255) break;   // inf / NaN: it stays what it is as far as "< 1e20" goes
        ++n_passes;
        const int es = e0 != 0 ? e0 : 1;
        const uint32_t I = e0 != 0 ? ((sbits & 0x7fffffu) | 0x800000u) : (sbits & 0x7fffffu);
        uint32_t a[8], y[8];
        uint32_t mine = 0;
        bool any_tie = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int sh = es - ex[j];
            const int shc = sh < 0 ? 0 : (sh > 25 ? 25 : sh);
            const unsigned long long w = (static_cast<unsigned long long>(M[j]) << 32) >> shc;
            y[j] = static_cast<uint32_t>(w);
            a[j] = sh < 0 ? (1u << 24) : static_cast<uint32_t>(w >> 32);   // x above the sum's binade: the crossing entry for sure
            mine += a[j] + (y[j] > 0x80000000u ? 1u : 0u);
            any_tie = any_tie || y[j] == 0x80000000u;
        }
        uint32_t pin = 0, s1 = mine, p0 = mine & 1u, p1 = p0 ^ 1u;
        const unsigned long long tie_lanes = __ballot(any_tie);
        if (tie_lanes != 0ull) {   // uniform
            if (any_tie) {
                uint32_t s0 = 0;
                s1 = 0, p0 = 0, p1 = 1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t alpha = a[j] & 1u, ab = y[j] > 0x80000000u ? 1u : 0u, ti = y[j] == 0x80000000u ? 1u : 0u;
                    s0 += a[j] + (ab | (ti & (p0 ^ alpha)));
                    s1 += a[j] + (ab | (ti & (p1 ^ alpha)));
                    p0 = ti ? 0u : (p0 ^ alpha ^ ab);
                    p1 = ti ? 0u : (p1 ^ alpha ^ ab);
                }
                mine = s0;
            }
            // the parity in front of this lane: behind the last lane that holds a tie (its outgoing parity is absolute), or
            // from I itself, times the xor of the lanes in between
            const unsigned long long valm = __ballot(p0 == 1u);
            const unsigned long long cm = tie_lanes & lt;
            if (cm == 0ull) {
                pin = (I & 1u) ^ (static_cast<uint32_t>(__popcll(valm & lt)) & 1u);
            } else {
                const int j = 63 - __clzll(static_cast<long long>(cm));
                const unsigned long long after = lt & ~((2ull << j) - 1ull);
                pin = (static_cast<uint32_t>(valm >> j) & 1u) ^ (static_cast<uint32_t>(__popcll(valm & after)) & 1u);
            }
            mine = pin ? s1 : mine;
        }
        mine = mine < (1u << 25) ? mine : (1u << 25);
        const uint32_t incl = wave_scan_u32(mine);
        const uint32_t limit = (1u << 24) - I;                  // I + increments reaching 2^24: the binade ends
        const unsigned long long crossm = __ballot(incl >= limit);
        if (crossm == 0ull) {
            const uint32_t In = I + static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));   // < 2^24
            s = __uint_as_float((e0 == 0 && In < (1u << 23)) ? In : ((static_cast<uint32_t>(es) << 23) | (In & 0x7fffffu)));
            break;
        }
        // the first lane whose entries reach 2^24 walks them; the crossing entry is added in fp32, and everything up to it
        // is done with (M = 0)
        const int lc = __builtin_ctzll(crossm);
        uint32_t run = I + (incl - mine), par = pin, hit = 8, run_at = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t alpha = a[j] & 1u, ab = y[j] > 0x80000000u ? 1u : 0u, ti = y[j] == 0x80000000u ? 1u : 0u;
            const uint32_t inc = a[j] + (ab | (ti & (par ^ alpha)));
            if (hit == 8 && run + inc >= (1u << 24)) {
                hit = j;
                run_at = run;
            }
            run += inc;
            par = ti ? 0u : (par ^ alpha ^ ab);
        }
        uint32_t xm = M[0];
        int xe = ex[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            xm = hit == static_cast<uint32_t>(j) ? M[j] : xm;
            xe = hit == static_cast<uint32_t>(j) ? ex[j] : xe;
        }
        const uint32_t hit_u = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(hit), lc));
        const uint32_t run_u = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(run_at), lc));   // < 2^24
        const uint32_t xm_u = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(xm), lc));
        const uint32_t xe_u = static_cast<uint32_t>(__builtin_amdgcn_readlane(xe, lc));
        // (M, ex) back to the float it came from: M carries the hidden bit exactly when the value is normal
        const float x_u = __uint_as_float(xm_u >= (1u << 23) ? ((xe_u << 23) | (xm_u & 0x7fffffu)) : xm_u);
        const float before = __uint_as_float((e0 == 0 && run_u < (1u << 23)) ? run_u : ((static_cast<uint32_t>(es) << 23) | (run_u & 0x7fffffu)));
        s = __fadd_rn(before, x_u);
        if (lane <= lc) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (lane < lc || static_cast<uint32_t>(j) <= hit) M[j] = 0u;
        }
    }
    return s;
}

// development aid (BYZ_BULYAN_CLOCKS=1): re-scores, their batches, integer passes and cycles
__device__ unsigned long long g_rescore_clock[14];

// One wave; wave-uniform result; `ok` = false when the row holds a negative value (the passes assume distances: the caller
// then takes the literal chain).  `head`: 512 floats of LDS for the literal chain over the first entries.
// `front` (LDS, one word per row of the workgroup, zero at the start of the loop): the number of leading 512-entry batches of
// this row that hold nothing but marks.  The winners of the picks so far are every central row's NEAREST neighbours, so the
// front of a contender's table fills with marks as the loop goes (up to ten batches at N = 10,000) -- and a mark never
// becomes live again, so a batch found empty once is skipped, unfetched, by every later re-score of the row (round 6).
// EXTRA (the speculative loop, bulyan_spec_kernel): `egone` is a bitmap over the POSITIONS of this row's table (LDS, the wave's own;
// bit p set: the entry at position p is gone although the table does not say so yet -- the winners of the earlier picks of a batch that
// is still being verified).  A lane's eight entries start at a multiple of eight: one byte of it.
template <bool CLOCKS, bool EXTRA = false>
__device__ __forceinline__ float reference_score_marked(const float* sorted_val, int n, int u, int take, int lane,
                                                        float* __restrict__ head, int head_chunks, bool& ok,
                                                        uint16_t* __restrict__ front, const uint8_t* __restrict__ egone = nullptr) {
    constexpr bool clocks = CLOCKS;   // (development: a compile-time switch -- what is compiled into this loop costs even when it never runs)
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const uint32_t* vals = reinterpret_cast<const uint32_t*>(sorted_val + static_cast<int64_t>(u) * n);
    const unsigned long long c_begin = clocks ? __builtin_readcyclecounter() : 0ull;
    unsigned long long n_passes = 0, n_batches = 0, c_passes = 0;
    struct Batch {
        uint32_t x[8];
    };
    auto fetch = [&](Batch& b, int r0) __attribute__((always_inline)) {
        int p = r0 + 8 * lane;
        p = p < n ? p : n;   // eight dwords from here stay inside the table's padding; what lies past the row is masked below
        const u32x4u lo = *reinterpret_cast<const u32x4u*>(vals + p), hi = *reinterpret_cast<const u32x4u*>(vals + p + 4);
        b.x[0] = lo.x, b.x[1] = lo.y, b.x[2] = lo.z, b.x[3] = lo.w;
        b.x[4] = hi.x, b.x[5] = hi.y, b.x[6] = hi.z, b.x[7] = hi.w;
    };
    float s = 0.0f;
    int got = 0, literal_left = head_chunks;
    bool negative = false;
    const int first_batch = front != nullptr ? __builtin_amdgcn_readfirstlane(static_cast<int>(*front)) : 0;   // (uniform)
    int empty_in_front = first_batch;   // batches known to hold only marks once this re-score is done
    bool in_front = true;
    // returns true behind the last batch of the prefix
    auto add = [&](const Batch& b, int r0) __attribute__((always_inline)) -> bool {
        uint32_t xb[8];
        uint32_t cnt = 0, top = 0;
        const int inside = n - r0 - 8 * lane;   // entries of this lane that lie inside the row (<= 0 .. >= 8)
        uint32_t eb = 0, lm = 0;                // (EXTRA) this lane's byte of the bitmap; its live entries
        if (EXTRA) {
            eb = inside > 0 ? egone[(r0 >> 3) + lane] : 0u;
            if (__ballot(eb != 0u) != 0ull) in_front = false;   // (a batch that is empty only on the bitmap's word is not remembered as empty)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = j < inside && b.x[j] != kGoneBits && !(EXTRA && ((eb >> j) & 1u) != 0u);
            xb[j] = live ? b.x[j] : 0u;
            cnt += live ? 1u : 0u;
            if (EXTRA) lm |= (live ? 1u : 0u) << j;
            top = xb[j] > top ? xb[j] : top;
        }
        negative = negative || top > kGoneBits;   // a sign bit on a live entry
        const uint32_t incl = wave_scan_u32(cnt);
        const int total = __builtin_amdgcn_readlane(static_cast<int>(incl), 63);
        if (total == 0) {
            // nothing live in these 512 entries: the winners of the picks so far are every row's NEAREST neighbours, so late in the
            // loop the front of the table is one run of marks -- up to ten such batches at N = 10,000 -- and there is nothing to add
            ++n_batches;
            if (in_front) ++empty_in_front;
            return r0 + 512 >= n;
        }
        in_front = false;
        if (got + total > take) {   // uniform: the prefix ends inside this batch
            int room = take - got - static_cast<int>(incl - cnt);   // live entries of this lane that still belong to it
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (EXTRA ? ((lm >> j) & 1u) != 0u : (xb[j] != 0u || (j < inside && b.x[j] == 0u))) {   // live (a live +0.0 counts)
                    if (room <= 0) xb[j] = 0u;   // past the prefix: + 0.0 (exact)
                    --room;
                }
            }
        }
        got += total;
        int start = 0;
        if (literal_left > 0) {
            // The first 64-entry chunks that hold anything go through the literal chain: the sum changes binade every other
            // step at first, then after 128, 256, ... entries, and up to ~512 entries a chain of additions is cheaper than a
            // pass and a crossing per binade.  Chunks of zeros (the attack's twins in front of a malicious row: hundreds of
            // +0.0) cost nothing and do not count.
            uint32_t any = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) any |= xb[j];
            const unsigned long long holds = __ballot(any != 0u);   // lanes 8 k .. 8 k + 7 hold chunk k
            if (holds != 0ull) {
                *reinterpret_cast<f32x4*>(head + 8 * lane) = f32x4{__uint_as_float(xb[0]), __uint_as_float(xb[1]), __uint_as_float(xb[2]), __uint_as_float(xb[3])};
                *reinterpret_cast<f32x4*>(head + 8 * lane + 4) = f32x4{__uint_as_float(xb[4]), __uint_as_float(xb[5]), __uint_as_float(xb[6]), __uint_as_float(xb[7])};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const f32x4* src = reinterpret_cast<const f32x4*>(head);
                for (int k = 0; k < 8 && literal_left > 0; ++k) {
                    if (((holds >> (8 * k)) & 0xffull) == 0ull) {
                        start = 64 * (k + 1);
                        continue;
                    }
                    f32x4 e[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) e[i] = src[16 * k + i];
#pragma unroll
                    for (int i = 0; i < 16; ++i) s = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(s, e[i].x), e[i].y), e[i].z), e[i].w);
                    --literal_left;
                    start = 64 * (k + 1);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // read before the next batch overwrites it
                __builtin_amdgcn_wave_barrier();
            } else {
                start = 512;
            }
        }
        if (__ballot(negative) == 0ull) {
            uint32_t M[8];
            int ex[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t e = xb[j] >> 23;
                ex[j] = e != 0u ? static_cast<int>(e) : 1;
                M[j] = e != 0u ? ((xb[j] & 0x7fffffu) | 0x800000u) : xb[j];
                if (8 * lane + j < start) M[j] = 0u;
            }
            const unsigned long long c0 = clocks ? __builtin_readcyclecounter() : 0ull;
            s = integer_passes(M, ex, s, lane, n_passes);
            if (clocks) c_passes += __builtin_readcyclecounter() - c0;
        }
        ++n_batches;
        return got >= take || r0 + 512 >= n;
    };
    Batch b0, b1, b2;
    const int r_first = 512 * first_batch;
    fetch(b0, r_first);
    fetch(b1, r_first + 512);
    fetch(b2, r_first + 1024);
    for (int r0 = r_first;; r0 += 3 * 512) {
        if (add(b0, r0)) break;
        fetch(b0, r0 + 3 * 512);
        if (add(b1, r0 + 512)) break;
        fetch(b1, r0 + 4 * 512);
        if (add(b2, r0 + 1024)) break;
        fetch(b2, r0 + 5 * 512);
    }
    ok = __ballot(negative) == 0ull;
    if (front != nullptr && lane == 0) *front = static_cast<uint16_t>(empty_in_front);
    if (clocks && lane == 0) {
        atomicAdd(&g_rescore_clock[0], 1ull);
        atomicAdd(&g_rescore_clock[1], n_batches);
        atomicAdd(&g_rescore_clock[2], n_passes);
        atomicAdd(&g_rescore_clock[3], __builtin_readcyclecounter() - c_begin);
        atomicAdd(&g_rescore_clock[4], c_passes);
    }
    return s;
}

struct GridDecision {
    int mode;        // 0 winner known, 1 round 2, 2 no candidate, 3 exchange timed out
    int winner;
    double threshold;
};

// (Round 4's exact incremental re-score -- a contender re-scored from a record of its previous chain, rows near the band kept
// scored -- was built, measured and REMOVED in round 5: EXPERIMENTS.md S2.)
template <bool DEV>
__global__ __launch_bounds__(kGridThreads) void bulyan_grid_kernel(
    const float* __restrict__ dist, int n, int theta, int drop, int users_count, int corrupted,
    const uint16_t* __restrict__ sorted_idx, const uint16_t* __restrict__ rank_t, float* sorted_val,
    const double* __restrict__ row_total, const double* __restrict__ row_top, const int32_t* __restrict__ cls,
    unsigned long long* __restrict__ xchg, float band_scale, int32_t* __restrict__ selection,
    int32_t* __restrict__ status, int32_t* __restrict__ rescored, int rescore_mode, int head_chunks, int skip_front) {
    __shared__ __attribute__((aligned(16))) float rescore_stage[kGridThreads / 64][512];
    __shared__ Candidate slots[kGridThreads / 64];
    __shared__ double second_slots[kGridThreads / 64];
    __shared__ uint32_t removed[kMaxSelectRows / 32];
    __shared__ GridDecision decision;
    __shared__ unsigned long long class_leader[256];
    __shared__ int leaders[kGridThreads];
    __shared__ int n_leaders;
    __shared__ uint16_t front_batches[kGridThreads];   // per local row: leading batches of its table that hold only marks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_wgs = gridDim.x, wg = blockIdx.x;
    const int u = wg * kGridThreads + tid;
    for (int i = tid; i < kMaxSelectRows / 32; i += kGridThreads) removed[i] = 0u;
    front_batches[tid] = 0;

    bool alive = u < n;
    double tot = alive ? row_total[u] : 0.0;
    double top = (alive && drop > 0) ? row_top[u] : 0.0;
    int bad = alive ? static_cast<int>(row_top[n + u]) : 0;   // live non-finite distances of this row
    int ptr = n - 1 - drop;
    const int my_class = alive ? cls[u] : 0;
    const int my_pos = visit_position(u);
    int n_rescored = 0;
    // rescore_mode >= 1: the table of ascending values carries the removals (-0.0); first of all the row's own diagonal
    const bool marked = rescore_mode >= 1;
    if (marked && alive) sorted_val[static_cast<int64_t>(u) * n + rank_t[static_cast<int64_t>(u) * n + u]] = __uint_as_float(kGoneBits);
    auto take_at = [&](int t) __attribute__((always_inline)) -> int {
        // the prefix the reference sums at pick t: sorted(...)[: users_count - t - f] of the n - t - 1 live entries
        const int live_entries = n - t - 1;
        const int keep = users_count - t - corrupted;
        return keep >= 0 ? (keep < live_entries ? keep : live_entries) : (live_entries + keep > 0 ? live_entries + keep : 0);
    };
    __syncthreads();

    // exchange slots: [parity][kind A, B, R][workgroup]
    auto slot = [&](int parity, int kind) { return xchg + (parity * 3 + kind) * kGridMaxWgs; };
    const unsigned long long none_a = (static_cast<unsigned long long>(kInfBits) << 32) | (static_cast<unsigned long long>(kNoRow) << 18);

    int result = 0;
    for (int t = 0; t < theta; ++t) {
        // granule A is rewritten at every pick, so a 4-bit tag tells pick t from pick t - 2 in the same slot; B and the
        // round-2 granule R (written only at picks that need it) carry the pick number itself
        const uint32_t tag = static_cast<uint32_t>((t >> 1) & 7) + 1u;
        const uint32_t tag18 = static_cast<uint32_t>(t + 1);
        const int parity = t & 1;
        const int take = take_at(t);
        // ---- 1. the workgroup's best row, and its best score outside that row's twin class
        const double score = tot - top;
        // (a non-finite distance inside the summed prefix makes the reference's score inf / NaN: never below 1e20)
        const bool candidate = alive && bad <= drop && score < static_cast<double>(kKrumInit);   // false for NaN
        Candidate c{static_cast<double>(kKrumInit), 0x7fffffff, -1, -1};
        if (candidate) c = Candidate{score, my_pos, u, my_class};
        const Candidate best = block_best(c, slots);
        const int best_class = best.cls;      // (travels with the winner: a load of cls[best.row] here cost every pick a round trip)
        double second = (candidate && my_class != best_class) ? score : __builtin_inf();
        second = wave_min_d(second);
        if (lane == 0) second_slots[wave] = second;
        __syncthreads();
        if (wave == 0) {
            // ---- 2. publish, gather, decide (all 64 lanes of wave 0 compute the same decision)
            double sec = second_slots[0];
#pragma unroll
            for (int w = 1; w < kGridThreads / 64; ++w) sec = fmin(sec, second_slots[w]);
            const float a_lb = best.row >= 0 ? float_below(best.score) : __builtin_inff();
            const float b_lb = float_below(sec);
            unsigned long long ga = best.row >= 0
                ? (static_cast<unsigned long long>(__float_as_uint(a_lb)) << 32) | (static_cast<unsigned long long>(best.row) << 18) |
                  (static_cast<unsigned long long>(best_class) << 4)
                : none_a;
            unsigned long long gb = static_cast<unsigned long long>(__float_as_uint(b_lb)) << 32;
            bool ok = true;
            if (n_wgs > 1) {
                if (lane == 0) {
                    granule_store(slot(parity, 0) + wg, ga | tag);
                    granule_store(slot(parity, 1) + wg, gb | tag18);
                }
                unsigned long long va = none_a, vb = static_cast<unsigned long long>(kInfBits) << 32;
                ok = gather_granule_pair(slot(parity, 0), slot(parity, 1), n_wgs, tag, tag18, lane, va, vb);
                ga = va;
                gb = vb;
            } else if (lane != 0) {
                ga = none_a;
                gb = static_cast<unsigned long long>(kInfBits) << 32;
            }
            // lane l now holds workgroup l's granules
            const float a = __uint_as_float(static_cast<uint32_t>(ga >> 32));
            const float b = __uint_as_float(static_cast<uint32_t>(gb >> 32));
            const int a_row = static_cast<int>((ga >> 18) & 0x3fffu);
            const int a_cls = static_cast<int>((ga >> 4) & 0x3fffu);
            const float m1 = wave_min_f(a);
            GridDecision d{0, -1, 0.0};
            if (!ok) {
                d.mode = 3;
            } else if (!(m1 < __builtin_inff())) {
                d.mode = 2;
            } else {
                // every row whose exact score lies within the band of the smallest one may be the reference's winner
                // band_scale >= 0: the rigorous bound 2.2 delta, delta = u (m + 1) / 2 (x band_scale);
                // band_scale <  0: |band_scale| u sqrt(m + 1), the random-walk size of the same error (not a bound)
                const double u24 = 5.9604644775390625e-08;
                const double band = (band_scale >= 0.0f ? static_cast<double>(band_scale) * 1.1 * u24 * static_cast<double>(take + 1)
                                                        : -static_cast<double>(band_scale) * u24 * sqrt(static_cast<double>(take + 1))) + 1e-9;
                const double ub = double_above(m1);
                const double thr = ub + fabs(ub) * (take <= 1 ? 1e-12 : band);
                d.threshold = thr;
                const bool in_a = static_cast<double>(a) <= thr;
                const bool in_b = static_cast<double>(b) <= thr;
                const unsigned long long first = __ballot(a == m1);
                const int lead_cls = __builtin_amdgcn_readlane(a_cls, __builtin_ctzll(first));
                const bool one_class = __ballot(in_b) == 0ull && __ballot(in_a && a_cls != lead_cls) == 0ull;
                if (one_class) {
                    const int pos = wave_min_i(in_a ? visit_position(a_row) : 0x7fffffff);
                    d.winner = pos == 0 ? 1 : (pos == 1 ? 0 : pos);   // visit_position is its own inverse
                } else {
                    d.mode = 1;
                }
            }
            if (lane == 0) decision = d;
        }
        if (tid < 256) class_leader[tid] = ~0ull;
        if (tid == 0) n_leaders = 0;
        __syncthreads();
        GridDecision d = decision;
        Candidate r{static_cast<double>(kKrumInit), 0x7fffffff, -1};
        if (d.mode == 1) {
            // ---- 3. round 2: the contenders scored in the reference's own arithmetic
            const bool contender = candidate && score <= d.threshold;
            const bool tracked = contender;
            const unsigned long long key = (static_cast<unsigned long long>(my_pos) << 32) | static_cast<uint32_t>(my_class);
            if (tracked) atomicMin(&class_leader[my_class & 255], key);
            __syncthreads();
            bool leader = false;
            if (tracked) {
                const unsigned long long held = class_leader[my_class & 255];
                // the class's earliest local member scores for the class; a class that lost its slot to another one
                // (hash collision) scores every member: redundant, never wrong
                leader = static_cast<int>(held & 0xffffffffu) != my_class || held == key;
            }
            if (leader) {
                const int at = atomicAdd(&n_leaders, 1);
                leaders[at] = tid;
            }
            __syncthreads();
            const int n_lead = n_leaders;
            for (int k = wave; k < n_lead; k += kGridThreads / 64) {
                const int row = __builtin_amdgcn_readfirstlane(wg * kGridThreads + leaders[k]);
                float s32 = 0.0f;
                bool done = false;
                if (marked) s32 = reference_score_marked<DEV>(sorted_val, n, row, take, lane, rescore_stage[wave], head_chunks, done,
                                                              skip_front != 0 ? &front_batches[leaders[k]] : nullptr);
                if (!done) s32 = reference_score_plain(sorted_val, sorted_idx, removed, n, row, take, lane, rescore_stage[wave]);
                if (s32 < kKrumInit) {
                    Candidate o{static_cast<double>(s32), visit_position(row), row};
                    if (better(o, r)) r = o;
                }
            }
        }
        if (d.mode == 1) {
            const int n_lead = n_leaders;
            if (wave == 0 && lane == 0) n_rescored += n_lead;
            const Candidate local = block_best(r, slots);   // every lane of a wave holds the same r
            if (wave == 0) {
                unsigned long long gr = local.row >= 0
                    ? (static_cast<unsigned long long>(__float_as_uint(static_cast<float>(local.score))) << 32) |
                      (static_cast<unsigned long long>(local.row) << 18)
                    : none_a;
                bool ok = true;
                if (n_wgs > 1) {
                    if (lane == 0) granule_store(slot(parity, 2) + wg, gr | tag18);
                    unsigned long long vr = none_a;
                    ok = gather_granules(slot(parity, 2), n_wgs, 0x3ffffu, tag18, lane, none_a, vr);
                    gr = vr;
                } else if (lane != 0) {
                    gr = none_a;
                }
                const int r_row = static_cast<int>((gr >> 18) & 0x3fffu);
                Candidate g{static_cast<double>(kKrumInit), 0x7fffffff, -1};
                if (r_row != static_cast<int>(kNoRow))
                    g = Candidate{static_cast<double>(__uint_as_float(static_cast<uint32_t>(gr >> 32))), visit_position(r_row), r_row};
                g = wave_best(g);
                if (lane == 0) {
                    decision.mode = !ok ? 3 : (g.row < 0 ? 2 : 0);
                    decision.winner = g.row;
                }
            }
            __syncthreads();
            d = decision;
        }
        if (d.mode != 0 || (n < 2)) {
            result = (n < 2) ? 1 : (d.mode == 3 ? 2 : 1);
            break;   // uniform across the grid: every workgroup reaches the same decision (or times out)
        }
        const int w = d.winner;
        if (tid == 0) {
            if (wg == 0) selection[t] = w;
            removed[w >> 5] |= 1u << (w & 31);
        }
        __syncthreads();
        // ---- 4. remove the winner from every row still present
        if (alive) {
            if (u == w) {
                alive = false;
            } else {
                const float dwf = dist[static_cast<int64_t>(w) * n + u];   // symmetric: d[w][u] == d[u][w]
                const bool dw_finite = __builtin_fabsf(dwf) <= 3.4028234663852886e38f;
                const double dw = dw_finite ? static_cast<double>(dwf) : 0.0;
                if (!dw_finite) --bad;
                const int r = rank_t[static_cast<int64_t>(w) * n + u];                          // rank of column w inside row u
                if (marked) sorted_val[static_cast<int64_t>(u) * n + r] = __uint_as_float(kGoneBits);
                tot -= dw;
                if (drop > 0 && r >= ptr) {
                    // w was one of this row's `drop` largest: the largest survivor below the boundary joins them
                    top -= dw;
                    int p = ptr - 1;
                    const uint16_t* order = sorted_idx + static_cast<int64_t>(u) * n;
                    while (p >= 0) {
                        const int col = order[p];
                        if (!((removed[col >> 5] >> (col & 31)) & 1u)) break;
                        --p;
                    }
                    if (p >= 0) {
                        const float joins = dist[static_cast<int64_t>(u) * n + order[p]];
                        if (__builtin_fabsf(joins) <= 3.4028234663852886e38f) top += static_cast<double>(joins);
                    }
                    ptr = p;
                }
            }
        }
    }
    if (tid == 0) {
        if (result != 0) atomicMax(status, result);
        if (n_rescored) atomicAdd(rescored, n_rescored);
    }
}


// ---------------------------------------------------------------------------------------------------
// The same loop, SPECULATIVE (round 6).  bulyan_grid_kernel spends two thirds of a contested pick waiting for ONE wave per
// contender to form the reference's sequential fp32 sum, and most picks are contested -- yet the exact (fp64) minimum is the
// reference's winner in all but a handful of picks per loop.  So a BATCH of up to 16 picks is decided optimistically: a contested
// pick takes the row with the smallest exact score at once and only remembers who its contenders were.  Behind the batch every
// (contested pick, contender) pair is scored the reference's way IN PARALLEL -- all four waves of every workgroup, pairs of different
// picks side by side -- in the state of ITS pick: the table of ascending values carries the marks of the picks before the batch, a
// bitmap over the row's positions (the wave's own, in LDS) those of the batch's earlier winners.  One exchange carries the
// workgroups' best contender of every contested pick; every workgroup finds the reference's winner of each and the first pick
// whose optimistic winner was not it.  None: the batch is committed (selection, marks).  Otherwise every row goes back to its
// snapshot of the batch's start, replays the picks in front of the wrong one (no decisions: the O(1) updates), takes the
// reference's winner there, commits that much, and the next batch starts behind it.  Every contested pick that is committed was
// decided by the reference's arithmetic in the reference's state, every other one by the proof of the band: the selection is
// bulyan_grid_kernel's, pick for pick.
constexpr int kSpecMax = 32;                       // picks per batch at most (the masks below are 32 bits wide)
constexpr int kSpecTable = 512;                    // (pick, twin class) -> leader, hashed

// Which row a slot (workgroup g, thread i) of the speculative loop owns.  The rows that are not the root of their twin class (the
// attack's copies: one class, a quarter of the rows) fill the slots from the front, in index order -- a class is scored once per workgroup
// that holds members, so it should sit in few.  All other rows are dealt over the slots behind them thread index by thread index,
// workgroup by workgroup, in the order of their first scores (T - Top, ties by index): the rows that contend are neighbours in that order.
// Slots without a row hold n.
// A workgroup = 64 rows x 4 waves: wave p counts a quarter of every tile of 256 rows for all 64 (round 6's last session: one
// thread per row walked all n rows alone -- 0.44 ms at N = 4000 on 16 CUs; the counts are integers, any split gives the same).
__global__ __launch_bounds__(kGridThreads) void spec_owner_kernel(const double* __restrict__ row_total, const double* __restrict__ row_top,
                                                                  const int32_t* __restrict__ cls, int n, int drop, int n_wgs,
                                                                  int32_t* __restrict__ owner) {
    static_assert(kGridThreads == 256, "four waves per workgroup");
    __shared__ double tile[kGridThreads];
    __shared__ int copy[kGridThreads];
    __shared__ int sums[3][64];
    const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
    const int u = blockIdx.x * 64 + lane;
    auto first_score = [&](int v) -> double {
        if (v >= n) return __builtin_inf();
        const double s = row_total[v] - (drop > 0 ? row_top[v] : 0.0);
        return s == s ? s : __builtin_inf();
    };
    const double su = first_score(u);
    const bool u_copy = u < n && cls[u] != u;
    int rank = 0, copies = 0, copies_before = 0;
    if (tid < 64) sums[0][tid] = sums[1][tid] = sums[2][tid] = 0;
    for (int v0 = 0; v0 < n; v0 += kGridThreads) {
        tile[tid] = first_score(v0 + tid);
        copy[tid] = (v0 + tid < n && cls[v0 + tid] != v0 + tid) ? 1 : 0;
        __syncthreads();
        const int m = n - v0 < kGridThreads ? n - v0 : kGridThreads;
        const int j_end = 64 * part + 64 < m ? 64 * part + 64 : m;
#pragma unroll 8
        for (int j = 64 * part; j < j_end; ++j) {
            const double sv = tile[j];
            const int cj = copy[j];
            copies += cj;
            copies_before += (cj != 0 && v0 + j < u) ? 1 : 0;
            rank += (cj == 0 && (sv < su || (sv == su && v0 + j < u))) ? 1 : 0;   // among the rows that are dealt
        }
        __syncthreads();
    }
    atomicAdd(&sums[0][lane], rank);
    atomicAdd(&sums[1][lane], copies);
    atomicAdd(&sums[2][lane], copies_before);
    __syncthreads();
    if (part != 0 || u >= n) return;
    rank = sums[0][lane];
    copies = sums[1][lane];
    copies_before = sums[2][lane];
    if (u_copy) {
        owner[copies_before] = u;
        return;
    }
    // the copies hold workgroups 0 .. q - 1 and threads 0 .. rem - 1 of workgroup q
    const int q = copies / kGridThreads, rem = copies % kGridThreads;
    const int narrow = n_wgs - q - 1, wide = n_wgs - q;   // workgroups with thread i free: i < rem, i >= rem
    const int first_part = rem * narrow;
    int wg, local;
    if (rank < first_part) {
        local = rank / narrow;
        wg = q + 1 + rank % narrow;
    } else {
        const int r = rank - first_part;
        local = rem + r / wide;
        wg = q + r % wide;
    }
    owner[wg * kGridThreads + local] = u;
}

__global__ __launch_bounds__(kGridThreads) void spec_identity_kernel(int n, int32_t* __restrict__ owner) {
    const int u = blockIdx.x * kGridThreads + threadIdx.x;
    if (u < n) owner[u] = u;
}

struct SpecVerdict {
    int k_bad;      // first pick of the batch whose optimistic winner is not the reference's (-1: none)
    int winner;     // the reference's winner there (-1: no row scores below 1e20)
    int timeout;
};

// kSpecThreads: 256 (the four waves that own the 256 rows) or 512 (four more that only score pairs and gather: from 6000 rows, where the
// verification is what a batch waits for -- N = 10,000: 68 -> 61 ms; they cost every pick 0.6 us at its barriers -- N = 4000: 12.6 -> 13.1 ms).
template <int kSpecThreads>
__global__ __launch_bounds__(kSpecThreads) void bulyan_spec_kernel(
    const float* __restrict__ dist, int n, int theta, int drop, int users_count, int corrupted,
    const uint16_t* __restrict__ sorted_idx, const uint16_t* __restrict__ rank_t, float* sorted_val,
    const double* __restrict__ row_total, const double* __restrict__ row_top, const int32_t* __restrict__ cls,
    unsigned long long* __restrict__ xchg, float band_scale, int32_t* __restrict__ selection,
    int32_t* __restrict__ status, int32_t* __restrict__ rescored, int head_chunks, int skip_front, int batch_picks,
    int32_t* __restrict__ spec_stats, const int32_t* __restrict__ owner) {
    __shared__ __attribute__((aligned(16))) float rescore_stage[kSpecThreads / 64][512];
    __shared__ Candidate slots[kSpecThreads / 64];
    __shared__ double second_slots[kSpecThreads / 64];
    __shared__ uint32_t removed[kMaxSelectRows / 32];
    __shared__ __attribute__((aligned(16))) uint32_t egone[kSpecThreads / 64][kMaxSelectRows / 32];   // per wave: see reference_score_marked<EXTRA>
    __shared__ GridDecision decision;
    __shared__ int decision_contested;
    __shared__ unsigned long long leader_of[kSpecTable];
    __shared__ uint16_t items[kGridThreads * kSpecMax];
    __shared__ int n_items;
    __shared__ Candidate wave_bests[kSpecThreads / 64][kSpecMax];
    __shared__ int winners[kSpecMax];
    __shared__ SpecVerdict verdict;
    __shared__ int true_row[kSpecMax];
    __shared__ uint16_t front_batches[kGridThreads];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_wgs = gridDim.x, wg = blockIdx.x;
    // threads 0 .. 255 own the workgroup's rows; the waves behind them only help: they score (pick, contender) pairs and gather
    // (which row a slot owns: spec_owner_kernel deals the rows to the workgroups in the order of their first scores, so that the rows
    // that contend -- the central ones, neighbours in that order -- are spread over all workgroups)
    const int u = tid < kGridThreads ? owner[wg * kGridThreads + tid] : n;
    for (int i = tid; i < kMaxSelectRows / 32; i += kSpecThreads) {
        removed[i] = 0u;
#pragma unroll
        for (int w = 0; w < kSpecThreads / 64; ++w) egone[w][i] = 0u;
    }
    if (tid < kGridThreads) front_batches[tid] = 0;

    bool alive = u < n;
    double tot = alive ? row_total[u] : 0.0;
    double top = (alive && drop > 0) ? row_top[u] : 0.0;
    int bad = alive ? static_cast<int>(row_top[n + u]) : 0;   // live non-finite distances of this row
    int ptr = n - 1 - drop;
    const int my_class = alive ? cls[u] : 0;
    const int my_pos = visit_position(u);
    int n_rescored = 0, n_batches = 0, n_rollbacks = 0, n_wasted = 0;
    if (alive) sorted_val[static_cast<int64_t>(u) * n + rank_t[static_cast<int64_t>(u) * n + u]] = __uint_as_float(kGoneBits);
    auto take_at = [&](int t) __attribute__((always_inline)) -> int {
        const int live_entries = n - t - 1;
        const int keep = users_count - t - corrupted;
        return keep >= 0 ? (keep < live_entries ? keep : live_entries) : (live_entries + keep > 0 ? live_entries + keep : 0);
    };
    __syncthreads();

    // exchange slots: [parity][kind A, B, (R of bulyan_grid_kernel: unused)][workgroup], then [parity][pick of the batch][workgroup]
    auto slot = [&](int parity, int kind) { return xchg + (parity * 3 + kind) * kGridMaxWgs; };
    auto batch_slot = [&](int parity, int k) { return xchg + (6 + parity * kSpecMax + k) * kGridMaxWgs; };
    const unsigned long long none_a = (static_cast<unsigned long long>(kInfBits) << 32) | (static_cast<unsigned long long>(kNoRow) << 18);

    // step 4 of bulyan_grid_kernel without the mark: the winner leaves every row's sums
    auto remove_winner = [&](int w) __attribute__((always_inline)) {
        if (tid == 0) removed[w >> 5] |= 1u << (w & 31);
        __syncthreads();
        if (alive) {
            if (u == w) {
                alive = false;
            } else {
                const float dwf = dist[static_cast<int64_t>(w) * n + u];   // symmetric: d[w][u] == d[u][w]
                const bool dw_finite = __builtin_fabsf(dwf) <= 3.4028234663852886e38f;
                const double dw = dw_finite ? static_cast<double>(dwf) : 0.0;
                if (!dw_finite) --bad;
                const int r = rank_t[static_cast<int64_t>(w) * n + u];                          // rank of column w inside row u
                tot -= dw;
                if (drop > 0 && r >= ptr) {
                    top -= dw;
                    int p = ptr - 1;
                    const uint16_t* order = sorted_idx + static_cast<int64_t>(u) * n;
                    while (p >= 0) {
                        const int col = order[p];
                        if (!((removed[col >> 5] >> (col & 31)) & 1u)) break;
                        --p;
                    }
                    if (p >= 0) {
                        const float joins = dist[static_cast<int64_t>(u) * n + order[p]];
                        if (__builtin_fabsf(joins) <= 3.4028234663852886e38f) top += static_cast<double>(joins);
                    }
                    ptr = p;
                }
            }
        }
    };

    int result = (n < 2 && theta > 0) ? 1 : 0;
    int t = 0;
    uint32_t seq = 0;        // decisions exchanged so far, discarded ones included: what the granules' tags count
    uint32_t batch_seq = 0;  // batches verified so far
    while (t < theta && result == 0) {
        // ---- the batch's start: every row's snapshot
        const bool s_alive = alive;
        const double s_tot = tot, s_top = top;
        const int s_bad = bad, s_ptr = ptr;
        const int t0 = t;
        const int kmax = theta - t0 < batch_picks ? theta - t0 : batch_picks;
        uint32_t cmask = 0;       // picks of this batch at which this row is a contender
        uint32_t contested = 0;   // (uniform) picks of this batch that more than one twin class contends for
        int n_done = 0, pending = 0;
        for (int k = 0; k < kmax; ++k, ++seq) {
            const int tt = t0 + k;
            const uint32_t tag = ((seq >> 1) & 7u) + 1u;
            const uint32_t tag18 = (seq + 1u) & 0x3ffffu;
            const int parity = static_cast<int>(seq & 1u);
            const int take = take_at(tt);
            // ---- 1. the workgroup's best row, and its best score outside that row's twin class (bulyan_grid_kernel)
            const double score = tot - top;
            const bool candidate = alive && bad <= drop && score < static_cast<double>(kKrumInit);   // false for NaN
            Candidate c{static_cast<double>(kKrumInit), 0x7fffffff, -1, -1};
            if (candidate) c = Candidate{score, my_pos, u, my_class};
            const Candidate best = block_best(c, slots);
            const int best_class = best.cls;
            double second = (candidate && my_class != best_class) ? score : __builtin_inf();
            second = wave_min_d(second);
            if (lane == 0) second_slots[wave] = second;
            __syncthreads();
            if (wave == 0) {
                // ---- 2. publish, gather, decide
                double sec = second_slots[0];
#pragma unroll
                for (int w = 1; w < kSpecThreads / 64; ++w) sec = fmin(sec, second_slots[w]);
                const float a_lb = best.row >= 0 ? float_below(best.score) : __builtin_inff();
                const float b_lb = float_below(sec);
                unsigned long long ga = best.row >= 0
                    ? (static_cast<unsigned long long>(__float_as_uint(a_lb)) << 32) | (static_cast<unsigned long long>(best.row) << 18) |
                      (static_cast<unsigned long long>(best_class) << 4)
                    : none_a;
                unsigned long long gb = static_cast<unsigned long long>(__float_as_uint(b_lb)) << 32;
                bool ok = true;
                if (n_wgs > 1) {
                    if (lane == 0) {
                        granule_store(slot(parity, 0) + wg, ga | tag);
                        granule_store(slot(parity, 1) + wg, gb | tag18);
                    }
                    unsigned long long va = none_a, vb = static_cast<unsigned long long>(kInfBits) << 32;
                    ok = gather_granule_pair(slot(parity, 0), slot(parity, 1), n_wgs, tag, tag18, lane, va, vb);
                    ga = va;
                    gb = vb;
                } else if (lane != 0) {
                    ga = none_a;
                    gb = static_cast<unsigned long long>(kInfBits) << 32;
                }
                const float a = __uint_as_float(static_cast<uint32_t>(ga >> 32));
                const float b = __uint_as_float(static_cast<uint32_t>(gb >> 32));
                const int a_row = static_cast<int>((ga >> 18) & 0x3fffu);
                const int a_cls = static_cast<int>((ga >> 4) & 0x3fffu);
                const float m1 = wave_min_f(a);
                GridDecision d{0, -1, 0.0};
                int is_contested = 0;
                if (!ok) {
                    d.mode = 3;
                } else if (!(m1 < __builtin_inff())) {
                    d.mode = 2;
                } else {
                    const double u24 = 5.9604644775390625e-08;
                    const double band = (band_scale >= 0.0f ? static_cast<double>(band_scale) * 1.1 * u24 * static_cast<double>(take + 1)
                                                            : -static_cast<double>(band_scale) * u24 * sqrt(static_cast<double>(take + 1))) + 1e-9;
                    const double ub = double_above(m1);
                    const double thr = ub + fabs(ub) * (take <= 1 ? 1e-12 : band);
                    d.threshold = thr;
                    const bool in_a = static_cast<double>(a) <= thr;
                    const bool in_b = static_cast<double>(b) <= thr;
                    const unsigned long long first = __ballot(a == m1);
                    const int lead_cls = __builtin_amdgcn_readlane(a_cls, __builtin_ctzll(first));
                    const bool one_class = __ballot(in_b) == 0ull && __ballot(in_a && a_cls != lead_cls) == 0ull;
                    // one class in the band: its earliest member, proven.  Otherwise, OPTIMISTICALLY, the smallest exact score
                    // (the earliest of the workgroups whose best rounds down to the same float): verified behind the batch
                    const int pos = wave_min_i((one_class ? in_a : a == m1) ? visit_position(a_row) : 0x7fffffff);
                    d.winner = pos == 0 ? 1 : (pos == 1 ? 0 : pos);   // visit_position is its own inverse
                    is_contested = one_class ? 0 : 1;
                }
                if (lane == 0) {
                    decision = d;
                    decision_contested = is_contested;
                }
            }
            __syncthreads();
            const GridDecision d = decision;
            if (d.mode == 3) {
                result = 2;
                break;
            }
            if (d.mode == 2) {   // no row scores below 1e20 at this pick -- if the picks before it stand
                pending = 1;
                ++seq;   // (this decision's granules are in the slots: the next decision must not take them for its own)
                break;
            }
            if (decision_contested != 0) {
                contested |= 1u << k;
                if (candidate && score <= d.threshold) cmask |= 1u << k;
            }
            if (tid == 0) winners[k] = d.winner;
            remove_winner(d.winner);
            ++n_done;
        }
        if (result != 0) break;
        contested &= n_done >= 32 ? ~0u : (1u << n_done) - 1u;
        cmask &= contested;

        // ---- 3. every (contested pick, contender) pair of the batch in the reference's arithmetic, in the state of its pick
        int k_bad = -1, w_true = -1;
        if (contested != 0u) {
            for (int i = tid; i < kSpecTable; i += kSpecThreads) leader_of[i] = ~0ull;
            if (tid == 0) n_items = 0;
            if (tid < (kSpecThreads / 64) * kSpecMax)
                (&wave_bests[0][0])[tid] = Candidate{static_cast<double>(kKrumInit), 0x7fffffff, -1, -1};
            __syncthreads();
            // one contender per (pick, twin class) and workgroup: the class's earliest local member (twins score alike)
            for (uint32_t m = cmask; m != 0u; m &= m - 1u) {
                const int k = __builtin_ctz(m);
                const unsigned long long key = (static_cast<unsigned long long>(my_pos) << 32) | (static_cast<uint32_t>(k) << 16) | static_cast<uint32_t>(my_class);
                atomicMin(&leader_of[(static_cast<uint32_t>(my_class) * 32u + static_cast<uint32_t>(k)) & (kSpecTable - 1)], key);
            }
            __syncthreads();
            for (uint32_t m = cmask; m != 0u; m &= m - 1u) {
                const int k = __builtin_ctz(m);
                const uint32_t low = (static_cast<uint32_t>(k) << 16) | static_cast<uint32_t>(my_class);
                const unsigned long long key = (static_cast<unsigned long long>(my_pos) << 32) | low;
                const unsigned long long held = leader_of[(static_cast<uint32_t>(my_class) * 32u + static_cast<uint32_t>(k)) & (kSpecTable - 1)];
                // a (pick, class) that lost its slot to another one scores every member: redundant, never wrong
                if (static_cast<uint32_t>(held) != low || held == key) items[atomicAdd(&n_items, 1)] = static_cast<uint16_t>((k << 8) | tid);
            }
            __syncthreads();
            const int n_it = n_items;
            uint32_t* const my_bits = egone[wave];
            for (int i = wave; i < n_it; i += kSpecThreads / 64) {
                const int it = __builtin_amdgcn_readfirstlane(static_cast<int>(items[i]));
                const int k = it >> 8;
                const int lt = it & 255;
                const int row = __builtin_amdgcn_readfirstlane(owner[wg * kGridThreads + lt]);
                // the batch's earlier winners are gone from this row in the state of pick k: their positions, on the wave's bitmap
                int pos_gone = -1;
                if (lane < k) {
                    pos_gone = rank_t[static_cast<int64_t>(winners[lane]) * n + row];
                    atomicOr(&my_bits[pos_gone >> 5], 1u << (pos_gone & 31));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                bool done = false;
                float s32 = reference_score_marked<false, true>(sorted_val, n, row, take_at(t0 + k), lane, rescore_stage[wave], head_chunks, done,
                                                                skip_front != 0 ? &front_batches[lt] : nullptr,
                                                                reinterpret_cast<const uint8_t*>(my_bits));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (pos_gone >= 0) my_bits[pos_gone >> 5] = 0u;
                if (!done) {
                    // (a sign bit on a live entry: the literal chain, with liveness from a bitmap of the columns in the state of pick k)
                    for (int j = lane; j < kMaxSelectRows / 32; j += 64) my_bits[j] = removed[j];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (lane >= k && lane < n_done) {
                        const int w = winners[lane];
                        atomicAnd(&my_bits[w >> 5], ~(1u << (w & 31)));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    s32 = reference_score_plain(sorted_val, sorted_idx, my_bits, n, row, take_at(t0 + k), lane, rescore_stage[wave]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    for (int j = lane; j < kMaxSelectRows / 32; j += 64) my_bits[j] = 0u;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (s32 < kKrumInit && lane == 0) {
                    const Candidate o{static_cast<double>(s32), visit_position(row), row, 0};
                    if (better(o, wave_bests[wave][k])) wave_bests[wave][k] = o;
                }
            }
            __syncthreads();
            {
                const uint32_t btag = (batch_seq + 1u) & 0x3ffffu;
                const int bpar = static_cast<int>(batch_seq & 1u);
                if (wave == 0) {
                    if (lane == 0) n_rescored += n_it;
                    // lane k: the workgroup's best contender of pick k, published for everybody
                    const bool mine_k = lane < n_done && ((contested >> lane) & 1u) != 0u;
                    Candidate local{static_cast<double>(kKrumInit), 0x7fffffff, -1, -1};
                    if (mine_k) {
#pragma unroll
                        for (int w = 0; w < kSpecThreads / 64; ++w)
                            if (better(wave_bests[w][lane], local)) local = wave_bests[w][lane];
                    }
                    if (n_wgs > 1) {
                        const unsigned long long gr = local.row >= 0
                            ? (static_cast<unsigned long long>(__float_as_uint(static_cast<float>(local.score))) << 32) |
                              (static_cast<unsigned long long>(local.row) << 18)
                            : none_a;
                        if (mine_k) granule_store(batch_slot(bpar, lane) + wg, gr | btag);
                    } else if (mine_k) {
                        true_row[lane] = local.row;
                    }
                }
                if (n_wgs > 1) {
                    // wave w gathers the picks w, w + 4, ...: the reference's winner of each (-1: none below 1e20; -2: timed out)
                    for (int k = wave; k < n_done; k += kSpecThreads / 64) {
                        if (((contested >> k) & 1u) == 0u) continue;
                        unsigned long long mine = none_a;
                        const bool ok = gather_granules(batch_slot(bpar, k), n_wgs, 0x3ffffu, btag, lane, none_a, mine);
                        const int r_row = static_cast<int>((mine >> 18) & 0x3fffu);
                        Candidate g{static_cast<double>(kKrumInit), 0x7fffffff, -1, -1};
                        if (r_row != static_cast<int>(kNoRow))
                            g = Candidate{static_cast<double>(__uint_as_float(static_cast<uint32_t>(mine >> 32))), visit_position(r_row), r_row, 0};
                        g = wave_best(g);
                        if (lane == 0) true_row[k] = ok ? g.row : -2;
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    SpecVerdict v{-1, -1, 0};
                    for (int k = 0; k < n_done && v.k_bad < 0 && v.timeout == 0; ++k) {
                        if (((contested >> k) & 1u) == 0u) continue;
                        if (true_row[k] == -2) {
                            v.timeout = 1;
                        } else if (true_row[k] != winners[k]) {
                            v.k_bad = k;
                            v.winner = true_row[k];
                        }
                    }
                    verdict = v;
                }
                __syncthreads();
            }
            ++batch_seq;
            if (verdict.timeout != 0) {
                result = 2;
                break;
            }
            k_bad = verdict.k_bad;
            w_true = verdict.winner;
        }
        if (k_bad >= 0) {
            // ---- back to the snapshot; the picks in front of the wrong one again (their winners stand), then the reference's winner
            ++n_rollbacks;
            n_wasted += n_done - k_bad;
            alive = s_alive, tot = s_tot, top = s_top, bad = s_bad, ptr = s_ptr;
            __syncthreads();
            if (tid == 0) {
                for (int k = 0; k < n_done; ++k) removed[winners[k] >> 5] &= ~(1u << (winners[k] & 31));
                if (w_true >= 0) winners[k_bad] = w_true;
            }
            __syncthreads();
            n_done = w_true >= 0 ? k_bad + 1 : k_bad;
            pending = w_true >= 0 ? 0 : 1;
            for (int k = 0; k < n_done; ++k) remove_winner(winners[k]);
        }
        // ---- commit: the selection, and the winners' marks in the rows that go on
        __syncthreads();
        for (int k = 0; k < n_done; ++k) {
            const int w = winners[k];
            if (wg == 0 && tid == 0) selection[t0 + k] = w;
            if (alive) sorted_val[static_cast<int64_t>(u) * n + rank_t[static_cast<int64_t>(w) * n + u]] = __uint_as_float(kGoneBits);
        }
        t = t0 + n_done;
        ++n_batches;
        if (pending != 0) result = 1;
        __syncthreads();   // (winners[] is rewritten by the next batch)
    }
    if (tid == 0) {
        if (result != 0) atomicMax(status, result);
        if (n_rescored) atomicAdd(rescored, n_rescored);
        if (wg == 0 && spec_stats != nullptr) {
            spec_stats[0] = n_batches;
            spec_stats[1] = n_rollbacks;
            spec_stats[2] = n_wasted;
        }
    }
}

}  // namespace

int64_t select_max_rows() { return kLargeMaxRows; }

// beyond the LDS-resident row sort and the 64 workgroups of the grid loop: large_rows.hip (BYZ_SELECT_LARGE=1: at every size, the tests)
bool select_large_applies(int64_t n) {
    if (n > kMaxSelectRows) return true;
    const char* e = std::getenv("BYZ_SELECT_LARGE");     // (read per call: the tests flip it inside one process)
    return e != nullptr && std::atoi(e) != 0;
}

int launch_row_sort(byz_ctx* ctx, const float* dist, int64_t n, int64_t prefix_len, int64_t drop_count,
                    bool want_tables, hipStream_t stream) {
    BYZ_REQUIRE(dist && n > 0, "row sort: bad arguments");
    if (n > kLargeMaxRows) {
        set_error("selection kernels support at most %lld rows, got %lld", (long long)kLargeMaxRows, (long long)n);
        return BYZ_E_UNSUPPORTED;
    }
    if (select_large_applies(n)) return launch_row_sort_large(ctx, dist, n, prefix_len, drop_count, want_tables, stream);
    int64_t n_pad = next_pow2(n);
    if (n_pad < 128) n_pad = 128;
    int threads = static_cast<int>(n_pad / 2);
    if (threads > 1024) threads = 1024;
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(n) * sizeof(float)));
    if (want_tables) {
        BYZ_TRY(ctx->sorted_idx.ensure(static_cast<size_t>(n) * n * sizeof(uint16_t)));
        BYZ_TRY(ctx->rank_t.ensure(static_cast<size_t>(n) * n * sizeof(uint16_t)));
        BYZ_TRY(ctx->rank_rows.ensure(static_cast<size_t>(n) * n * sizeof(uint16_t)));
        BYZ_TRY(ctx->row_total.ensure(static_cast<size_t>(n) * sizeof(double)));
        BYZ_TRY(ctx->row_top.ensure(static_cast<size_t>(2 * n) * sizeof(double)));   // sums, then the counts of non-finite entries
        BYZ_TRY(ctx->sorted_val.ensure(static_cast<size_t>(n) * n * sizeof(float) + 64));   // (+ 64: a re-score reads 8 dwords per lane)
    }
    // BYZ_ROW_SORT_BLOCKED=0: the textbook network (one level per pass over LDS) for every size -- the same-box A/B and the
    // bitwise comparison (tests/test_gpu_round4.py); default: the register-blocked network from 256 keys
    const char* e_blocked = std::getenv("BYZ_ROW_SORT_BLOCKED");     // (read per call: the tests flip it inside one process)
    const bool blocked = (e_blocked == nullptr || std::atoi(e_blocked) != 0) && n_pad >= 256;
    // the blocked network keeps 16 keys per thread: one thread per work item, so that a CU holds several rows at once below
    // 16,384 keys (4 at 4096: the network's compare-exchange chains are latency, and four waves per row leave a CU idle)
    if (blocked) threads = static_cast<int>(n_pad / 16 < 64 ? 64 : n_pad / 16);
    const size_t key_words = blocked ? static_cast<size_t>(n_pad + (n_pad >> 4)) : static_cast<size_t>(n_pad);
    const size_t lds = key_words * 8 + static_cast<size_t>(threads) * 8;
    KernelTimer t(ctx, BYZ_K_ROW_SORT, stream);
#define BYZ_ROW_SORT(TB, BL, ...)                                                                                       \
    do {                                                                                                                \
        BYZ_HIP(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&row_sort_kernel<TB, BL>), static_cast<int>(lds))); \
        row_sort_kernel<TB, BL><<<static_cast<unsigned>(n), threads, lds, stream>>>(                                     \
            dist, (int)n, (int)n_pad, (int)prefix_len, (int)drop_count, ctx->scores.as<float>(), __VA_ARGS__);           \
    } while (0)
    if (want_tables) {
        if (blocked)
            BYZ_ROW_SORT(true, true, ctx->sorted_idx.as<uint16_t>(), ctx->rank_rows.as<uint16_t>(), ctx->row_total.as<double>(),
                         ctx->row_top.as<double>(), ctx->sorted_val.as<float>());
        else
            BYZ_ROW_SORT(true, false, ctx->sorted_idx.as<uint16_t>(), ctx->rank_rows.as<uint16_t>(), ctx->row_total.as<double>(),
                         ctx->row_top.as<double>(), ctx->sorted_val.as<float>());
        BYZ_TRY(check_launch("row_sort_kernel"));
        const unsigned tiles = static_cast<unsigned>(ceil_div(n, 64));
        rank_transpose_kernel<<<dim3(tiles, tiles), 256, 0, stream>>>(ctx->rank_rows.as<uint16_t>(), (int)n, ctx->rank_t.as<uint16_t>());
        return check_launch("rank_transpose_kernel");
    } else {
        if (blocked) BYZ_ROW_SORT(false, true, nullptr, nullptr, nullptr, nullptr, nullptr);
        else BYZ_ROW_SORT(false, false, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
#undef BYZ_ROW_SORT
    return check_launch("row_sort_kernel");
}

int launch_krum_argmin(byz_ctx* ctx, int64_t n, int32_t* winner_dev, hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_KRUM_ARGMIN, stream);
    krum_argmin_kernel<<<1, 1024, 0, stream>>>(ctx->scores.as<float>(), (int)n, winner_dev);
    return check_launch("krum_argmin_kernel");
}

int launch_bulyan_loop(byz_ctx* ctx, const float* dist, int64_t n, int64_t theta, int64_t drop_count,
                       int64_t users_count, int64_t corrupted, int32_t* selection_dev, int32_t* status_dev,
                       hipStream_t stream) {
    BYZ_REQUIRE(dist && selection_dev && status_dev && n > 0 && theta >= 0 && theta <= n,
                "bulyan loop: bad arguments (n=%lld theta=%lld)", (long long)n, (long long)theta);
    BYZ_TRY(ctx->twin_class.ensure(static_cast<size_t>(2 * n) * sizeof(int32_t)));
    // granules: [2][A, B, R][64 workgroups], then the speculative loop's [2][32 picks of a batch][64], then its three counters
    constexpr size_t kGranules = static_cast<size_t>(2 * 3 + 2 * kSpecMax) * kGridMaxWgs;
    BYZ_TRY(ctx->xchg.ensure((kGranules + 2) * sizeof(unsigned long long)));
    int32_t* cls_tmp = ctx->twin_class.as<int32_t>();
    int32_t* cls = cls_tmp + n;
    // which arithmetic decides a pick whose contenders lie within rounding of each other (see bulyan_grid_kernel):
    //   BYZ_BULYAN_BAND unset / "rigorous"  every row that CAN beat the minimum in sequential fp32 is re-scored
    //   BYZ_BULYAN_BAND=<x> (x > 0)          the rigorous band scaled by x; (x < 0) |x| u sqrt(m): statistical, not a bound
    float band_scale = 1.0f;
    if (const char* e = std::getenv("BYZ_BULYAN_BAND")) {
        if (std::strcmp(e, "rigorous") != 0) band_scale = static_cast<float>(std::atof(e));
    }
    // BYZ_BULYAN_RESCORE=plain: the literal chain of additions with liveness from the bitmap (round 2: the form the C oracle
    // was checked against); default: the marked table and the integer passes -- the same bits
    int rescore_mode = 1;
    if (const char* e = std::getenv("BYZ_BULYAN_RESCORE")) rescore_mode = std::strcmp(e, "plain") == 0 ? 0 : 1;
    // 64-entry chunks of a re-score that go through the literal chain before the passes take over (measured, N = 4000 / 10,000
    // scaled / 10,000 attack: 1 chunk 32.7 / 163 / 95.7 ms, 4 chunks 30.9 / 158 / 95.8, 8 chunks 30.4 / 157 / 94.1)
    const int head_chunks = 8;
    // BYZ_BULYAN_FRONT=0: every re-score starts at the row's first entry (round 5's behaviour: the same-box A/B)
    int skip_front = 1;
    if (const char* e = std::getenv("BYZ_BULYAN_FRONT")) skip_front = std::atoi(e) != 0 ? 1 : 0;
    const char* clocks_env = std::getenv("BYZ_BULYAN_CLOCKS");
    const bool clocks = rescore_mode != 0 && clocks_env != nullptr && std::atoi(clocks_env) != 0;
    if (clocks) {
        rescore_mode += 1;
        const unsigned long long zero[14] = {0};
        BYZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_rescore_clock), zero, sizeof(zero)));
    }
    BYZ_HIP(hipMemsetAsync(ctx->xchg.ptr, 0, (kGranules + 2) * sizeof(unsigned long long), stream));
    BYZ_HIP(hipMemsetAsync(status_dev, 0, 3 * sizeof(int32_t), stream));   // status, rows re-scored, (unused)
    KernelTimer t(ctx, BYZ_K_BULYAN_LOOP, stream);
    twin_class_kernel<<<static_cast<unsigned>(ceil_div(n, 4)), 256, 0, stream>>>(dist, (int)n, cls_tmp);
    BYZ_TRY(check_launch("twin_class_kernel"));
    twin_class_fix_kernel<<<static_cast<unsigned>(ceil_div(n, 256)), 256, 0, stream>>>(cls_tmp, (int)n, cls);
    BYZ_TRY(check_launch("twin_class_fix_kernel"));
    if (select_large_applies(n))      // beyond 16,384 rows (large_rows.hip, inside this launcher's timer); the twin classes serve it too
        return launch_bulyan_loop_large(ctx, dist, n, theta, drop_count, users_count, corrupted, cls, selection_dev, status_dev, stream);
    const unsigned n_wgs = static_cast<unsigned>(ceil_div(n, kGridThreads));   // <= 64: all resident, they wait for each other
    // BYZ_BULYAN_BATCH=<k>: picks decided optimistically before their contested ones are verified together (bulyan_spec_kernel;
    // default 32 from 1000 rows, 16 from 6000; at most 32); 0: bulyan_grid_kernel, every contested pick re-scored before the next one (rounds 2-5; also taken for
    // BYZ_BULYAN_RESCORE=plain and BYZ_BULYAN_CLOCKS).  The same selection, pick for pick.
    // (measured, same box, N = 4000 / N = 10,000 on hard data: batches of 8: 15.8 / 101.0 ms, 16: 14.0 / 102.2, 24: 13.55 / 103.4, 32: 13.47 / 103.5;
    // a batch that grows behind a batch that stood and halves behind a roll-back: 13.7 / 102.6 -- not kept.  Below ~1000 rows few picks
    // are contested and the batches' bookkeeping costs more than it saves: N = 300: 0.49 -> 0.58 ms, N = 700: 1.40 -> 1.52; N = 1000: 2.18 -> 2.08)
    // (with eight waves per workgroup, the bench's data: N = 4000: 24 -> 13.4 ms, 32 -> 13.3; configs[4]'s slice, N = 10,000: 12 -> 70.0, 16 -> 71.2,
    // 24 -> 73.4, 32 -> 77.1 ms: a wrong pick costs the picks decided behind it, and there are more of them at N = 10,000)
    int batch = n >= 6000 ? 16 : (n >= 1000 ? 32 : 0);
    if (const char* e = std::getenv("BYZ_BULYAN_BATCH")) batch = std::atoi(e);
    if (batch > kSpecMax) batch = kSpecMax;
    if (batch >= 1 && rescore_mode == 1 && !clocks) {
        int32_t* stats = reinterpret_cast<int32_t*>(ctx->xchg.as<unsigned long long>() + kGranules);
        BYZ_TRY(ctx->spec_owner.ensure(static_cast<size_t>(n_wgs) * kGridThreads * sizeof(int32_t)));
        int32_t* owner = ctx->spec_owner.as<int32_t>();
        BYZ_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(owner), static_cast<int>(n), static_cast<size_t>(n_wgs) * kGridThreads, stream));
        // BYZ_BULYAN_DEAL=0: slot (g, i) owns row 256 g + i (the comparison)
        const char* deal_env = std::getenv("BYZ_BULYAN_DEAL");
        if (deal_env == nullptr || std::atoi(deal_env) != 0) {
            spec_owner_kernel<<<static_cast<unsigned>(ceil_div(n, 64)), kGridThreads, 0, stream>>>(
                ctx->row_total.as<double>(), ctx->row_top.as<double>(), cls, (int)n, (int)drop_count, (int)n_wgs, owner);
        } else {
            spec_identity_kernel<<<n_wgs, kGridThreads, 0, stream>>>((int)n, owner);
        }
        BYZ_TRY(check_launch("spec_owner_kernel"));
        auto* spec = n >= 6000 ? &bulyan_spec_kernel<512> : &bulyan_spec_kernel<256>;
        spec<<<n_wgs, n >= 6000 ? 512 : 256, 0, stream>>>(
            dist, (int)n, (int)theta, (int)drop_count, (int)users_count, (int)corrupted, ctx->sorted_idx.as<uint16_t>(),
            ctx->rank_t.as<uint16_t>(), ctx->sorted_val.as<float>(), ctx->row_total.as<double>(), ctx->row_top.as<double>(), cls,
            ctx->xchg.as<unsigned long long>(), band_scale, selection_dev, status_dev, status_dev + 1, head_chunks, skip_front, batch, stats,
            owner);
        BYZ_TRY(check_launch("bulyan_spec_kernel"));
        if (const char* e = std::getenv("BYZ_BULYAN_STATS"); e != nullptr && std::atoi(e) != 0) {
            int32_t host[4] = {0, 0, 0, 0};
            BYZ_HIP(hipStreamSynchronize(stream));
            BYZ_HIP(hipMemcpy(host, stats, 3 * sizeof(int32_t), hipMemcpyDeviceToHost));
            BYZ_HIP(hipMemcpy(host + 3, status_dev + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
            std::fprintf(stderr, "bulyan (speculative, batches of %d): %d picks in %d batches, %d rolled back (%d picks decided again), %d re-scores\n",
                         batch, (int)theta, host[0], host[1], host[2], host[3]);
        }
        return BYZ_OK;
    }
    auto* kernel = clocks ? &bulyan_grid_kernel<true> : &bulyan_grid_kernel<false>;
    kernel<<<n_wgs, kGridThreads, 0, stream>>>(
        dist, (int)n, (int)theta, (int)drop_count, (int)users_count, (int)corrupted, ctx->sorted_idx.as<uint16_t>(),
        ctx->rank_t.as<uint16_t>(), ctx->sorted_val.as<float>(), ctx->row_total.as<double>(), ctx->row_top.as<double>(), cls,
        ctx->xchg.as<unsigned long long>(), band_scale, selection_dev, status_dev, status_dev + 1, rescore_mode, head_chunks,
        skip_front);
    BYZ_TRY(check_launch("bulyan_grid_kernel"));
    if (clocks) {
        unsigned long long c[14];
        BYZ_HIP(hipStreamSynchronize(stream));
        BYZ_HIP(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_rescore_clock), sizeof(c)));
        const double r = c[0] ? static_cast<double>(c[0]) : 1.0;
        std::fprintf(stderr, "bulyan re-scores: %llu; per re-score: %.1f batches, %.1f integer passes, %.0f cycles (%.0f inside the passes)\n",
                     c[0], c[1] / r, c[2] / r, c[3] / r, c[4] / r);
    }
    return BYZ_OK;
}

}  // namespace byz
