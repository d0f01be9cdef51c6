// Median-window trimmed mean by counting selection, reference defences.py:44-52, per parameter (column):
//
//   med  = np.median(column)                       fp32; even count -> (a + b) / 2
//   good = sorted(column - med, key=abs)[:k]       stable: ties in |x - med| keep the lower row first
//   out  = np.mean(good) + med
//
// The reference sorts every column; nothing in the result needs the full order.  Only three order
// statistics matter: the median (one or two middle ranks) and t = the k-th smallest |x - med|.  Given t,
//     sum(good) = sum of d over {|d| < t}  +  (k - #{|d| < t}) tied values of magnitude exactly t,
// and the row order only matters when +t and -t both occur and not all of them are kept.
//
// An order statistic of R values held in registers costs ONE v_cmp per value per probe:
// count(T) = #{x <= T} is a ballot + s_bcnt1 per register, accumulated on the scalar unit.  Probes bisect
// the value range (arithmetic midpoints, falling back to midpoints of the order-preserving integer keys so
// that at most 12 + 32 probes are ever made) until at most 64 candidates remain inside the bracket; those
// are compacted into one value per lane (ballot + mbcnt scatter through a 256-byte LDS strip) and sorted
// by a 64-lane bitonic network.  About 35 VALU operations per matrix element in total, against ~200 for a
// bitonic sort of 1024 values: the kernel sits near the HBM time instead of 6x above it.
//
// Data movement (gfx950): one workgroup = 8 waves = one tile of 32 consecutive columns x all rows.
//   * global loads are 128-byte row segments (8 lanes x dwordx4), 4 in flight per thread, software
//     pipelined against the LDS transit of the previous 256-row chunk;
//   * the LDS transit buffer (256 rows x 36 floats, 36 KB) only transposes: wave w takes the 4 columns
//     4w..4w+3, lane l takes rows l, l+64, ... with conflict-free ds_read_b128, so a lane ends up with
//     RPL = ceil(R/64) values of each of its 4 columns in registers.  The tile lives in the register file
//     (512 KB per CU), which is what lets R reach 2560 rows; LDS (160 KB) could not hold it;
//   * after staging, a wave never synchronises with another wave again: its 4 columns are entirely its own.
// Rows past R are padded with quiet NaNs: they fail every ordered comparison, so no counting pass, min/max
// or sum ever sees them.  A NaN in the data itself makes np.median, and with it the reference's result, NaN.
//
// Algorithmic traffic: 4 bytes read per (row, column), 4 bytes written per column.  Bound: HBM.
#include "common.hpp"

#include "lane_exchange.hpp"

namespace byz {
namespace {

using namespace lanes;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kCols = 32;                 // columns per tile
constexpr int kStride = 36;               // floats per LDS row: 16-byte aligned, b128 column reads conflict-free
constexpr int kThreads = 512;             // 8 waves x 4 columns
constexpr int kWaves = kThreads / 64;
constexpr int kMaxRpl = 40;               // 2560 rows
constexpr int kArithProbes = 12;          // value-space bisection steps before switching to key space

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

// number of values <= T (NaN padding never counts).  GS registers per uniform guard.
template <int RPL, bool ABS>
__device__ __forceinline__ int count_le(const float (&v)[RPL], int groups, float T) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;
    int c = 0;
#pragma unroll
    for (int g = 0; g < RPL / GS; ++g) {
        if (g < groups) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) c += __popcll(__ballot(mag<ABS>(v[g * GS + jj]) <= T));
        }
    }
    return c;
}

// wave-uniform values are moved to the scalar file explicitly: the bisection control flow is scalar code
__device__ __forceinline__ float uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
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

struct Pair {
    float a, b;
};

// The r-th smallest (0-based) of mag(v), and with WANT_NEXT also the (r+1)-th.
// Requires on entry: count(lo) == 0, count(hi) == c_hi > r (+1 with WANT_NEXT).
template <int RPL, bool ABS, bool WANT_NEXT>
__device__ __forceinline__ Pair select_rank(const float (&v)[RPL], int groups, int r, float lo, float hi, int c_hi,
                                            int lane, float* strip) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;
    const int r_hi = WANT_NEXT ? r + 1 : r;
    int c_lo = 0;
    for (int it = 0; c_hi - c_lo > 64; ++it) {
        const uint32_t klo = fkey(lo), khi = fkey(hi);
        if (khi - klo <= 1u) return Pair{hi, hi};  // no float strictly inside: every candidate equals hi
        float T = 0.5f * lo + 0.5f * hi;
        const bool arith = it < kArithProbes && finite_f(lo) && finite_f(hi) && T > lo && T < hi;
        if (!arith) T = from_fkey(klo + ((khi - klo) >> 1));
        const int c = count_le<RPL, ABS>(v, groups, T);
        if (c <= r) {
            lo = T;
            c_lo = c;
        } else if (c > r_hi) {
            hi = T;
            c_hi = c;
        } else {
            // WANT_NEXT and c == r + 1: T separates the two wanted ranks
            float below = -__builtin_inff(), above = __builtin_inff();
#pragma unroll
            for (int g = 0; g < RPL / GS; ++g) {
                if (g < groups) {
#pragma unroll
                    for (int jj = 0; jj < GS; ++jj) {
                        const float a = mag<ABS>(v[g * GS + jj]);
                        below = (a <= T) ? __builtin_fmaxf(below, a) : below;
                        above = (a > T) ? __builtin_fminf(above, a) : above;
                    }
                }
            }
            return Pair{wave_max(below), wave_min(above)};
        }
    }
    // at most 64 candidates in (lo, hi]: one per lane, sorted across the wave
    int n_cand = 0;
#pragma unroll
    for (int g = 0; g < RPL / GS; ++g) {
        if (g < groups) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) {
                const float a = mag<ABS>(v[g * GS + jj]);
                const bool in = !(a <= lo) && a <= hi;   // lo may be NaN when the minimum is -inf
                const unsigned long long m = __ballot(in);
                if (m) {  // uniform
                    const int pos = n_cand + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32),
                                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
                    if (in) strip[pos] = a;
                    n_cand += __popcll(m);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float s[1][1];
    s[0][0] = lane < n_cand ? strip[lane] : __builtin_inff();
    wave_bitonic_sort<1, 1>(s, lane);
    __builtin_amdgcn_wave_barrier();  // the strip is reused by the next selection
    const int i = r - c_lo;
    Pair out;
    out.a = uniform(__shfl(s[0][0], i, 64));
    out.b = WANT_NEXT ? uniform(__shfl(s[0][0], i + 1, 64)) : out.a;
    return out;
}

// One column: everything after the values sit in registers.  Returns the reference's out[i].
template <int RPL>
__device__ __forceinline__ float median_window_column(float (&v)[RPL], int groups, int n, int keep, float mn, float mx,
                                                      int lane, float* strip) {
    constexpr int GS = RPL >= 4 ? 4 : RPL;
    // ---- median (defences.py:49)
    float med;
    {
        // lo: the float just below the minimum (count 0); hi: the maximum (count n)
        const float lo = from_fkey(fkey(mn) - 1u);
        if (n & 1) {
            med = select_rank<RPL, false, false>(v, groups, (n - 1) >> 1, lo, mx, n, lane, strip).a;
        } else {
            const Pair p = select_rank<RPL, false, true>(v, groups, (n >> 1) - 1, lo, mx, n, lane, strip);
            med = __fmul_rn(__fadd_rn(p.a, p.b), 0.5f);
        }
    }
    // ---- deviations in place (defences.py:50: column - med); rounding is monotone, so the largest
    // |deviation| belongs to one of the extremes
#pragma unroll
    for (int g = 0; g < RPL / GS; ++g) {
        if (g < groups) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) v[g * GS + jj] = __fsub_rn(v[g * GS + jj], med);
        }
    }
    const float max_dev = __builtin_fmaxf(__builtin_fabsf(__fsub_rn(mn, med)), __builtin_fabsf(__fsub_rn(mx, med)));
    // ---- t = the keep-th smallest |deviation|
    const float t = select_rank<RPL, true, false>(v, groups, keep - 1, -__uint_as_float(1u), max_dev, n, lane, strip).a;
    // ---- everything strictly closer than t is kept; of the ties at exactly t, the first `need` in row order
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
        // +t and -t both present and only some are kept: the reference's stable sort keeps the lowest rows
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
    // defences.py:51: np.mean(good) + med
    return __fadd_rn(__fdiv_rn(sum, static_cast<float>(keep)), med);
}

template <int RPL>
__global__ __launch_bounds__(kThreads, (RPL <= 16 ? 4 : 2)) void median_window_kernel(const float* __restrict__ G, int n_rows,
                                                                 int64_t n_cols, int64_t ld,
                                                                 const int32_t* __restrict__ row_index, int keep,
                                                                 float* __restrict__ out) {
    constexpr int JC = RPL >= 4 ? 4 : RPL;   // registers (64-row groups) per transit chunk == guard group
    constexpr int NCH = RPL / JC;
    __shared__ __attribute__((aligned(16))) float transit[64 * JC * kStride + kWaves * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* strip = transit + 64 * JC * kStride + wave * 64;
    const int64_t c_base = static_cast<int64_t>(blockIdx.x) * kCols;
    const float qnan = __uint_as_float(0x7fc00000u);
    const int chunks = (n_rows + 64 * JC - 1) / (64 * JC);   // == guard groups in use

    // ---- stage: global (128-byte row segments) -> LDS transit -> registers (4 columns x RPL rows per lane)
    const int ld_q = (tid & 7) * 4, ld_r = tid >> 3;
    const int64_t ld_c = c_base + ld_q;
    f32x4 tmp[JC];
    auto fetch = [&](int ch) {
#pragma unroll
        for (int p = 0; p < JC; ++p) {
            const int row = ch * 64 * JC + 64 * p + ld_r;
            f32x4 val = {qnan, qnan, qnan, qnan};
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
    float x[4][RPL];
    float mn[4], mx[4];
    int n_nan[4] = {0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mn[c] = __builtin_inff();
        mx[c] = -__builtin_inff();
    }
    fetch(0);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch < chunks) {
#pragma unroll
            for (int p = 0; p < JC; ++p)
                *reinterpret_cast<f32x4*>(transit + (64 * p + ld_r) * kStride + ld_q) = tmp[p];
            if (ch + 1 < chunks) fetch(ch + 1);
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < JC; ++jj) {
                const f32x4 val = *reinterpret_cast<const f32x4*>(transit + (64 * jj + lane) * kStride + 4 * wave);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float e = val[c];
                    x[c][ch * JC + jj] = e;
                    mn[c] = __builtin_fminf(mn[c], e);   // minnum / maxnum: NaN padding is ignored
                    mx[c] = __builtin_fmaxf(mx[c], e);
                    n_nan[c] += __popcll(__ballot(e != e));
                }
            }
            __syncthreads();
        }
    }
    const int pad_slots = chunks * 64 * JC - n_rows;

    // ---- per column: selection by counting, no further workgroup synchronisation
    float result[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float lo = wave_min(mn[c]), hi = wave_max(mx[c]);
        float r;
        if (keep <= 0 || n_nan[c] > pad_slots) {
            r = qnan;   // np.mean([]) is nan; a NaN in the column makes np.median nan
        } else {
            r = median_window_column<RPL>(x[c], chunks, n_rows, keep, lo, hi, lane, strip);
        }
        result[c] = r;
    }
    if (lane < 4) {
        const int64_t col = c_base + 4 * wave + lane;
        const float r = lane == 0 ? result[0] : (lane == 1 ? result[1] : (lane == 2 ? result[2] : result[3]));
        if (col < n_cols) out[col] = r;
    }
}

template <int RPL>
int launch_rpl(const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index, int64_t keep,
               float* out, hipStream_t stream) {
    const int64_t n_tiles = ceil_div(n_cols, kCols);
    median_window_kernel<RPL><<<static_cast<unsigned>(n_tiles), kThreads, 0, stream>>>(
        G, static_cast<int>(n_rows), n_cols, ld, row_index, static_cast<int>(keep), out);
    return check_launch("median_window_kernel");
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
    const int64_t rpl = ceil_div(n_rows, 64);
    if (rpl > kMaxRpl) return launch_trimmed_mean_sorted(ctx, G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 1) return launch_rpl<1>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 2) return launch_rpl<2>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 4) return launch_rpl<4>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 8) return launch_rpl<8>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 16) return launch_rpl<16>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 24) return launch_rpl<24>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    if (rpl <= 32) return launch_rpl<32>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
    return launch_rpl<40>(G, n_rows, n_cols, ld, row_index, keep, out, stream);
}

}  // namespace byz
