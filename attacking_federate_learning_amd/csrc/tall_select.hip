// Trimmed mean of TALL columns (more rows than the register kernels of median_window.hip hold: 5,632), reference defences.py:44-52.
//
//   med  = np.median(column)                       fp32; even count -> (a + b) / 2
//   good = sorted(column - med, key=abs)[:k]       stable: ties in |x - med| keep the lower row first
//   out  = np.mean(good) + med
//
// Nothing in the result needs the column's full order (median_window.hip): the median is one or two order statistics of the
// values, the window's edge t is ONE order statistic of |fl(x - med)|, and the mean is a sum over {|d| < t} plus the tied values at
// t in row order.  Until round 6's last session every height beyond 5,632 rows was SORTED (an LDS bitonic network up to 16,384 rows,
// a global-memory one beyond: 13.7 ms for 10,000 x 32,768 = 96 GB/s, six times the register kernels' cost per value).  Here an order
// statistic is a RADIX SELECT over the 32-bit keys, 8 bits per pass, the column streamed from HBM once per pass:
//   * a workgroup owns 64 consecutive columns; its four waves split the rows (wave w: rows w, w + 4, ...), a wave reads 256
//     contiguous bytes per row;
//   * a pass counts the keys that match the digits found so far into hist[digit][column] (LDS, one bank per column: no
//     conflicts whatever the data), then one thread per column walks its 256 counters to the digit that holds the rank;
//   * 4 passes give the lower median (and, on the way, the next larger value for an even count and whether a NaN is there), 4 the
//     window's edge, the last one the sums: 9 passes, 36 bytes per value -- bound by HBM like the sort never was.
// Ties at the edge with both signs present (+t and -t: the only place where row order matters) are settled by one thread per
// column walking the rows in order; a NaN anywhere in the column makes the result NaN, like np.median.
#include "common.hpp"

#include <cstdlib>

namespace byz {
namespace {

constexpr int kTileCols = 64;
constexpr int kWaves = 16;     // sixteen waves split the rows: what covers HBM's latency is loads in flight (four waves: 1.2 TB/s)
constexpr int kUnroll = 8;
// hist[256][64] (the waves' partial sums and counts live in it once the selects are done), then per column: the digits found so
// far, the rank left, and the sixteen 16-digit segment sums of the walk to the rank's digit
constexpr int kHistWords = 256 * kTileCols;
constexpr int kLdsWords = kHistWords + kTileCols * (2 + kWaves + 3);

__device__ __forceinline__ uint32_t ordered_bits(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

struct Column {
    const float* G;
    int64_t ld;
    const int32_t* row_index;
    int n_rows;
    int64_t col;      // this thread's column (clamped into the matrix; `real` says whether it exists)
    __device__ __forceinline__ float at(int r) const {
        const int64_t src = row_index ? row_index[r] : r;
        return G[src * ld + col];
    }
};

// STAGE 1: the value's order-preserving bits; STAGE 2: the bits of |fl(x - med)| (non-negative floats order like their bits)
template <int STAGE>
__device__ __forceinline__ uint32_t key_of(float x, float med) {
    if constexpr (STAGE == 1) return ordered_bits(x);
    else return __float_as_uint(__builtin_fabsf(__fsub_rn(x, med)));
}

// The key of rank `rank` (0-based, ascending) among the column's keys.  Every thread of the workgroup calls it; the result is
// the same in the four threads of a column.  `left_out` / `equal_out`: the rank inside the run of equal keys, and its length.
template <int STAGE>
__device__ __forceinline__ uint32_t radix_select(const Column& c, float med, int rank, uint32_t* hist, uint32_t* found, int* left,
                                                 uint32_t* seg, uint32_t* extra, int tx, int ty, int& left_out, int& equal_out) {
    static_assert(kWaves == 16, "the walk to the rank's digit takes sixteen 16-digit segments, one per wave");
    // STAGE 1 also leaves (for the median of an even count, and np.median's NaN) in extra[0..63] whether the column holds a NaN and in
    // extra[128..191] the smallest key above the answer: the first pass sees every value; the last pass knows the answer's upper 24 bits,
    // so a larger key either shares them (the next occupied digit of the last histogram) or lies beyond them (a running minimum)
    if (STAGE == 1 && ty == 0) {
        extra[tx] = 0u;
        extra[kTileCols + tx] = 0xffffffffu;
    }
    const int tid = ty * kTileCols + tx;
    if (ty == 0) {
        found[tx] = 0u;
        left[tx] = rank;
    }
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < kHistWords; i += kTileCols * kWaves) hist[i] = 0u;
        __syncthreads();
        const uint32_t prefix = found[tx];
        uint32_t beyond = 0xffffffffu;
        int nan = 0;
        for (int r0 = ty; r0 < c.n_rows; r0 += kWaves * kUnroll) {
            float v[kUnroll];
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const int r = r0 + j * kWaves;
                v[j] = c.at(r < c.n_rows ? r : c.n_rows - 1);
            }
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const int r = r0 + j * kWaves;
                const uint32_t key = key_of<STAGE>(v[j], med);
                const bool match = shift == 24 || (key >> (shift + 8)) == prefix;
                if (r < c.n_rows && match) atomicAdd(&hist[((key >> shift) & 255u) * kTileCols + tx], 1u);
                if constexpr (STAGE == 1) {      // (a clamped row repeats a real value: harmless for both)
                    if (shift == 24) nan |= (__float_as_uint(v[j]) & 0x7fffffffu) > 0x7f800000u ? 1 : 0;
                    if (shift == 0) beyond = (key >> 8) > prefix && key < beyond ? key : beyond;
                }
            }
        }
        if constexpr (STAGE == 1) {
            if (shift == 24 && nan != 0) atomicOr(&extra[tx], 1u);
            if (shift == 0 && beyond != 0xffffffffu) atomicMin(&extra[kTileCols + tx], beyond);
        }
        __syncthreads();
        {   // the digit that holds the rank: sixteen segment sums per column first, then one thread walks 16 + 16 counters
            uint32_t seg_sum = 0u;
#pragma unroll
            for (int b = 0; b < 16; ++b) seg_sum += hist[(16 * ty + b) * kTileCols + tx];
            seg[ty * kTileCols + tx] = seg_sum;
        }
        __syncthreads();
        if (ty == 0) {
            int want = left[tx];
            int s16 = 15;
            for (int g = 0; g < 16; ++g) {
                const int cnt = static_cast<int>(seg[g * kTileCols + tx]);
                if (want < cnt) {
                    s16 = g;
                    break;
                }
                want -= cnt;
            }
            int digit = 16 * s16 + 15;
            uint32_t count = 0u;
            for (int b = 16 * s16; b < 16 * s16 + 16; ++b) {
                count = hist[b * kTileCols + tx];
                if (want < static_cast<int>(count)) {
                    digit = b;
                    break;
                }
                want -= static_cast<int>(count);
            }
            found[tx] = (prefix << 8) | static_cast<uint32_t>(digit);
            left[tx] = want;
            if (shift == 0) {
                seg[tx] = count;      // (the run of keys equal to the answer)
                if constexpr (STAGE == 1) {
                    uint32_t next = extra[kTileCols + tx];
                    for (int b = digit + 1; b < 256; ++b) {
                        if (hist[b * kTileCols + tx] != 0u) {
                            next = (prefix << 8) | static_cast<uint32_t>(b);
                            break;
                        }
                    }
                    extra[2 * kTileCols + tx] = next;
                }
            }
        }
        __syncthreads();
    }
    left_out = left[tx];
    equal_out = static_cast<int>(seg[tx]);
    const uint32_t key = found[tx];
    __syncthreads();      // (found / left / hist are rewritten by the next call)
    return key;
}

__global__ __launch_bounds__(kTileCols * kWaves) void tall_select_kernel(const float* __restrict__ G, int n_rows, int64_t n_cols,
                                                                         int64_t ld, const int32_t* __restrict__ row_index, int keep,
                                                                         float* __restrict__ out) {
    extern __shared__ uint32_t lds[];
    uint32_t* const hist = lds;
    uint32_t* const found = lds + kHistWords;
    int* const left = reinterpret_cast<int*>(found + kTileCols);
    uint32_t* const seg = reinterpret_cast<uint32_t*>(left + kTileCols);               // [kWaves][64]
    uint32_t* const extra = seg + kWaves * kTileCols;                                   // [3][64]: NaN seen, minimum beyond, next key
    double* const part_sum = reinterpret_cast<double*>(hist);                           // [kWaves][64], over the dead histogram
    int* const part_less = reinterpret_cast<int*>(part_sum + kWaves * kTileCols);       // [kWaves][64]
    int* const part_pos = part_less + kWaves * kTileCols;
    int* const part_neg = part_pos + kWaves * kTileCols;
    const int tx = threadIdx.x & (kTileCols - 1), ty = threadIdx.x / kTileCols;
    const int64_t col = static_cast<int64_t>(blockIdx.x) * kTileCols + tx;
    const bool real = col < n_cols;
    const Column c{G, ld, row_index, n_rows, real ? col : n_cols - 1};

    // ---- the median: rank (n - 1) / 2, and for an even count the next value above it
    const int mid = (n_rows - 1) >> 1;
    int run_left, run_len;
    const uint32_t lower_key = radix_select<1>(c, 0.0f, mid, hist, found, left, seg, extra, tx, ty, run_left, run_len);
    float med = from_ordered_bits(lower_key);
    {
        const uint32_t above = extra[2 * kTileCols + tx];      // the smallest key above the lower median
        const int any_nan = static_cast<int>(extra[tx]);        // a NaN anywhere: np.median is NaN then
        if (any_nan) {
            if (ty == 0 && real) out[col] = __uint_as_float(0x7fc00000u);
            // (the column goes on through the passes with the others of its tile -- the barriers are the workgroup's -- and its
            //  result is not written again)
        }
        if ((n_rows & 1) == 0) {
            // ranks mid and mid + 1: the same value while the run of equal keys reaches beyond rank mid
            const float upper = run_left + 1 < run_len ? med : from_ordered_bits(above);
            med = __fmul_rn(__fadd_rn(med, upper), 0.5f);
        }
        if (keep <= 0) {      // np.mean([]) -> nan
            if (ty == 0 && real) out[col] = __uint_as_float(0x7fc00000u);
            return;
        }
        // ---- the window's edge: the keep-th smallest |fl(x - med)|
        int edge_left, edge_len;
        const uint32_t t_bits = radix_select<2>(c, med, keep - 1, hist, found, left, seg, extra, tx, ty, edge_left, edge_len);
        // ---- the sums: everything closer than the edge, and the tied values at it by sign
        double sum = 0.0;      // (np.mean sums pairwise: close to exact; a lane's share of a 2^20-row column is 262,144 values)
        int n_less = 0, n_pos = 0, n_neg = 0;
        for (int r0 = ty; r0 < n_rows; r0 += kWaves * kUnroll) {
            float v[kUnroll];
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const int r = r0 + j * kWaves;
                v[j] = c.at(r < n_rows ? r : n_rows - 1);
            }
#pragma unroll
            for (int j = 0; j < kUnroll; ++j) {
                const int r = r0 + j * kWaves;
                const float d = __fsub_rn(v[j], med);
                const uint32_t a = __float_as_uint(__builtin_fabsf(d));
                const bool in = r < n_rows;
                if (in && a < t_bits) {
                    sum += static_cast<double>(d);
                    ++n_less;
                }
                n_pos += in && a == t_bits && d > 0.0f ? 1 : 0;
                n_neg += in && a == t_bits && d < 0.0f ? 1 : 0;
            }
        }
        part_sum[ty * kTileCols + tx] = sum;
        part_less[ty * kTileCols + tx] = n_less;
        part_pos[ty * kTileCols + tx] = n_pos;
        part_neg[ty * kTileCols + tx] = n_neg;
        __syncthreads();
        if (ty != 0 || !real || any_nan) return;
        double total = 0.0;
        int less = 0, pos = 0, neg = 0;
        for (int w = 0; w < kWaves; ++w) {      // a fixed order: the same bits every run
            total += part_sum[w * kTileCols + tx];
            less += part_less[w * kTileCols + tx];
            pos += part_pos[w * kTileCols + tx];
            neg += part_neg[w * kTileCols + tx];
        }
        const float edge = __uint_as_float(t_bits);
        const int want = keep - less;            // tied values the window still takes (all of them unless the tie crosses the edge)
        if (want >= pos + neg) {
            total += static_cast<double>(pos - neg) * static_cast<double>(edge);
        } else if (neg == 0 || pos == 0) {
            total += static_cast<double>(pos ? want : -want) * static_cast<double>(edge);
        } else {
            // +t and -t both present and not all of them fit: the reference's stable sort takes them in ROW order (defences.py:50)
            int taken = 0, tp = 0, tn = 0;
            for (int r = 0; r < n_rows && taken < want; ++r) {
                const float d = __fsub_rn(c.at(r), med);
                if (__float_as_uint(__builtin_fabsf(d)) == t_bits && d != 0.0f) {
                    tp += d > 0.0f ? 1 : 0;
                    tn += d < 0.0f ? 1 : 0;
                    ++taken;
                }
            }
            total += static_cast<double>(tp - tn) * static_cast<double>(edge);
        }
        out[col] = __fadd_rn(static_cast<float>(total / static_cast<double>(keep)), med);
    }
}

}  // namespace

// BYZ_TM_TALL=0: the sorts of rounds 3-6 (trimmed_mean_lds_kernel up to 16,384 rows, the global-memory segment sort beyond)
bool trimmed_mean_tall_applies(int64_t n_rows) {
    if (n_rows <= 5632) return false;
    const char* e = std::getenv("BYZ_TM_TALL");     // (read per call: the tests flip it inside one process)
    return e == nullptr || std::atoi(e) != 0;
}

int launch_trimmed_mean_tall(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                             int64_t keep, float* out, hipStream_t stream) {
    if (n_rows > kLargeMaxRows) {
        set_error("trimmed_mean supports at most %lld rows, got %lld", (long long)kLargeMaxRows, (long long)n_rows);
        return BYZ_E_UNSUPPORTED;
    }
    const int64_t tiles = ceil_div(n_cols, kTileCols);
    BYZ_REQUIRE(tiles <= 0x7fffffff, "trimmed_mean: too many columns");
    const int lds_bytes = kLdsWords * 4;
    BYZ_HIP(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(&tall_select_kernel), lds_bytes));
    tall_select_kernel<<<static_cast<unsigned>(tiles), kTileCols * kWaves, lds_bytes, stream>>>(G, (int)n_rows, n_cols, ld, row_index,
                                                                                             (int)keep, out);
    return check_launch("tall_select_kernel");
}

}  // namespace byz
