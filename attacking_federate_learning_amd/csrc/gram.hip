// Pairwise client distances, reference defences.py:16-21 (_krum_create_distances).
//
// The reference evaluates ||g_i - g_j|| pair by pair; here the N x N matrix comes from the Gram identity
//     d_ij = sqrt(max(0, c_ii + c_jj - 2 c_ij)),   C = G * G^T,
// because C is a genuine dense contraction over the D parameters and that is what the matrix cores are
// for.  fp32 data is kept exact: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate), which is bit-for-bit a
// k-ordered fmaf chain.
//
// Data layout and tiling (gfx950):
//   * G is row-major (N x D): both operands of C = G G^T are K-contiguous, so every global load is a
//     128-byte row segment (BK = 32 floats) read as dwordx4 by 8 adjacent lanes.
//   * One workgroup = 4 waves = one 128 x 128 tile of C (lower triangle only, tj <= ti) over one K slice
//     (split-K): small N has too few tiles to fill 256 CUs, so the D axis supplies the parallelism.
//   * Operand tiles go global -> registers -> LDS (row stride 36 floats: ds_read_b128 of 16 different
//     rows at one k offset is conflict-free), double-buffered, one barrier per K stage; the next stage's
//     global loads are in flight while the current stage's MFMAs run.
//   * Each wave owns a 64 x 64 sub-tile = 2 x 2 MFMA 32x32 blocks = 64 accumulator registers.
//
// Numerics:
//   * an fp32 accumulator chain never exceeds kFlushK = 2048 products; longer K ranges are flushed into
//     fp64 running sums, and the split-K slabs are reduced in fp64 in a fixed order (deterministic);
//   * c_ii, c_jj and c_ij all come out of the same code with the same k order, so bitwise-identical rows
//     (every malicious client submits the same vector, malicious.py:26-27) give d_ij == 0 exactly and
//     identical distance rows -- the exact ties the reference resolves by visit order survive.
//
// Algorithmic work per call: N^2 * D flops (half Gram, 2 flops per MAC), 4 N D bytes read.  Bound: fp32
// MFMA (157.3 TF) once N/4 flop/B exceeds the machine balance (N >~ 80), HBM below that.
#include "common.hpp"

#include <cstdlib>

namespace byz {
namespace {

constexpr int TM = 128;              // tile edge (rows of G per operand tile)
constexpr int BK = 32;               // floats of K per stage
constexpr int LDS_STRIDE = BK + 4;   // 36 floats = 144 B: 16-byte aligned, conflict-free b128 reads
constexpr int THREADS = 256;
constexpr int kFlushK = 2048;        // longest fp32 accumulation chain
constexpr int TILE_FLOATS = TM * LDS_STRIDE;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void tile_coords(int t, int& ti, int& tj) {
    // t = ti*(ti+1)/2 + tj, 0 <= tj <= ti
    int r = static_cast<int>((sqrtf(8.0f * static_cast<float>(t) + 1.0f) - 1.0f) * 0.5f);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r;
    tj = t - r * (r + 1) / 2;
}

// 4 consecutive floats of one row, zero-filled outside [0, n_rows) x [k_lo, k_hi)
__device__ __forceinline__ f32x4 load_row_segment(const float* __restrict__ G, int64_t ld, int64_t n_rows,
                                                  int64_t row, int64_t k, int64_t k_hi) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row < n_rows) {
        const float* p = G + row * ld + k;
        if (k + 4 <= k_hi) {
            v = *reinterpret_cast<const f32x4u*>(p);
        } else {
            if (k + 0 < k_hi) v.x = p[0];
            if (k + 1 < k_hi) v.y = p[1];
            if (k + 2 < k_hi) v.z = p[2];
        }
    }
    return v;
}

template <typename PartialT>
__global__ __launch_bounds__(THREADS, 2) void gram_tile_kernel(const float* __restrict__ G, int64_t n_rows,
                                                               int64_t n_cols, int64_t ld,
                                                               int64_t stages_per_split,
                                                               PartialT* __restrict__ partial, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_FLOATS];  // [2 buffers][A | B][TM][LDS_STRIDE]

    const int tile = blockIdx.x;
    const int split = blockIdx.y;
    int ti, tj;
    tile_coords(tile, ti, tj);
    const bool diagonal = (ti == tj);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int64_t k_begin = static_cast<int64_t>(split) * stages_per_split * BK;
    int64_t k_end = k_begin + stages_per_split * BK;
    if (k_end > n_cols) k_end = n_cols;
    const int n_stages = k_begin < k_end ? static_cast<int>((k_end - k_begin + BK - 1) / BK) : 0;

    // staging assignment: 8 lanes cover one 128-byte row segment, 32 rows per pass, 4 passes per operand
    const int ld_kq = (tid & 7) * 4;
    const int ld_row = tid >> 3;
    const int64_t a_row0 = static_cast<int64_t>(ti) * TM;
    const int64_t b_row0 = static_cast<int64_t>(tj) * TM;

    f32x4 ra[4], rb[4];
    auto fetch = [&](int stage) {
        const int64_t k = k_begin + static_cast<int64_t>(stage) * BK + ld_kq;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = load_row_segment(G, ld, n_rows, a_row0 + ld_row + 32 * p, k, k_end);
            if (!diagonal) rb[p] = load_row_segment(G, ld, n_rows, b_row0 + ld_row + 32 * p, k, k_end);
        }
    };
    auto stash = [&](int buf) {
        float* A = lds + buf * 2 * TILE_FLOATS;
        float* B = A + TILE_FLOATS;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(A + (ld_row + 32 * p) * LDS_STRIDE + ld_kq) = ra[p];
            if (!diagonal) *reinterpret_cast<f32x4*>(B + (ld_row + 32 * p) * LDS_STRIDE + ld_kq) = rb[p];
        }
    };

    f32x16 acc[2][2];
    constexpr bool kWide = sizeof(PartialT) == 8;
    double wide[kWide ? 64 : 1];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;
    if constexpr (kWide) {
#pragma unroll
        for (int e = 0; e < 64; ++e) wide[e] = 0.0;
    }

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;
    constexpr int kFlushStages = kFlushK / BK;

    if (n_stages > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    for (int s = 0; s < n_stages; ++s) {
        if (s + 1 < n_stages) fetch(s + 1);
        const float* A = lds + (s & 1) * 2 * TILE_FLOATS;
        const float* B = diagonal ? A : A + TILE_FLOATS;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
                a[m] = *reinterpret_cast<const f32x4*>(A + (wr * 64 + m * 32 + frag_row) * LDS_STRIDE + kk * 8 + frag_k);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                b[n] = *reinterpret_cast<const f32x4*>(B + (wc * 64 + n * 32 + frag_row) * LDS_STRIDE + kk * 8 + frag_k);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
        }
        if (s + 1 < n_stages) stash((s + 1) & 1);
        __syncthreads();
        if constexpr (kWide) if ((s + 1) % kFlushStages == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        wide[(m * 2 + n) * 16 + e] += static_cast<double>(acc[m][n][e]);
                        acc[m][n][e] = 0.0f;
                    }
        }
    }

    PartialT* out = partial + (static_cast<int64_t>(split) * n_tiles + tile) * (TM * TM);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = wr * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int j = wc * 64 + n * 32 + (lane & 31);
                if constexpr (kWide)
                    out[i * TM + j] = static_cast<PartialT>(wide[(m * 2 + n) * 16 + e] + static_cast<double>(acc[m][n][e]));
                else
                    out[i * TM + j] = static_cast<PartialT>(acc[m][n][e]);
            }
}

// gram[i][j] = sum over splits (fixed order) of the slab entry of the lower-triangle tile holding (i, j).
template <typename PartialT>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const PartialT* __restrict__ partial, int n_tiles,
                                                          int splits, int64_t n, double* __restrict__ gram) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
    const int64_t i = static_cast<int64_t>(blockIdx.y) * 4 + (threadIdx.x >> 6);
    if (i >= n || j >= n) return;
    const int64_t hi = i > j ? i : j, lo = i > j ? j : i;
    const int ti = static_cast<int>(hi / TM), tj = static_cast<int>(lo / TM);
    const int tile = ti * (ti + 1) / 2 + tj;
    const int64_t off = static_cast<int64_t>(tile) * (TM * TM) + (hi % TM) * TM + (lo % TM);
    double s = 0.0;
    for (int sp = 0; sp < splits; ++sp)
        s += static_cast<double>(partial[static_cast<int64_t>(sp) * n_tiles * (TM * TM) + off]);
    gram[i * n + j] = s;
}

__global__ __launch_bounds__(256) void distance_kernel(const double* __restrict__ gram, int64_t n,
                                                       float* __restrict__ dist) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
    const int64_t i = static_cast<int64_t>(blockIdx.y) * 4 + (threadIdx.x >> 6);
    if (i >= n || j >= n) return;
    float d;
    if (i == j) {
        d = __builtin_inff();  // the reference keeps no self-distance (defences.py:18-20)
    } else {
        const double d2 = gram[i * n + i] + gram[j * n + j] - 2.0 * gram[i * n + j];
        // rounding can leave a tiny negative value for near-identical rows; NaN (poisoned input) must stay NaN
        d = static_cast<float>(sqrt(d2 < 0.0 ? 0.0 : d2));
    }
    dist[i * n + j] = d;
}

int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : fallback;
}

}  // namespace

int launch_gram(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, double* gram,
                hipStream_t stream) {
    BYZ_REQUIRE(G && gram && n_rows > 0 && n_cols > 0 && ld >= n_cols, "gram: bad shape %lld x %lld ld %lld",
                (long long)n_rows, (long long)n_cols, (long long)ld);
    const int64_t T = ceil_div(n_rows, TM);
    const int64_t n_tiles = T * (T + 1) / 2;
    if (n_tiles > 0x7fffffff) {
        set_error("gram: too many tiles");
        return BYZ_E_UNSUPPORTED;
    }
    const int64_t stages = ceil_div(n_cols, BK);
    // split-K: aim at ~3 workgroups per CU in total, but keep at least 16 stages (512 columns) per slab so
    // that slab traffic stays a small fraction of the matrix traffic
    int64_t splits = ceil_div(static_cast<int64_t>(ctx->num_cus) * 3, n_tiles);
    const int64_t min_stages = env_int("BYZ_GRAM_MIN_STAGES", 16);
    if (splits > ceil_div(stages, min_stages)) splits = ceil_div(stages, min_stages);
    const int forced = env_int("BYZ_GRAM_SPLITS", 0);
    if (forced > 0) splits = forced;
    if (splits < 1) splits = 1;
    if (splits > stages) splits = stages;
    if (splits > 65535) splits = 65535;
    const int64_t stages_per_split = ceil_div(stages, splits);
    splits = ceil_div(stages, stages_per_split);
    const bool wide = stages_per_split * BK > kFlushK;
    const size_t slab = static_cast<size_t>(TM) * TM * (wide ? sizeof(double) : sizeof(float));
    BYZ_TRY(ctx->gram_partials.ensure(static_cast<size_t>(splits) * n_tiles * slab));
    const size_t lds_bytes = 0;  // static LDS: 2 buffers x (A|B) x 128 x 36 floats = 73,728 B -> 2 workgroups per CU
    {
        KernelTimer t(ctx, BYZ_K_GRAM, stream);
        dim3 grid(static_cast<unsigned>(n_tiles), static_cast<unsigned>(splits));
        if (wide)
            gram_tile_kernel<double><<<grid, THREADS, lds_bytes, stream>>>(G, n_rows, n_cols, ld, stages_per_split, ctx->gram_partials.as<double>(), (int)n_tiles);
        else
            gram_tile_kernel<float><<<grid, THREADS, lds_bytes, stream>>>(G, n_rows, n_cols, ld, stages_per_split, ctx->gram_partials.as<float>(), (int)n_tiles);
        BYZ_TRY(check_launch("gram_tile_kernel"));
    }
    {
        KernelTimer t(ctx, BYZ_K_GRAM_REDUCE, stream);
        dim3 grid(static_cast<unsigned>(ceil_div(n_rows, 64)), static_cast<unsigned>(ceil_div(n_rows, 4)));
        if (wide)
            gram_reduce_kernel<double><<<grid, 256, 0, stream>>>(ctx->gram_partials.as<double>(), (int)n_tiles, (int)splits, n_rows, gram);
        else
            gram_reduce_kernel<float><<<grid, 256, 0, stream>>>(ctx->gram_partials.as<float>(), (int)n_tiles, (int)splits, n_rows, gram);
        BYZ_TRY(check_launch("gram_reduce_kernel"));
    }
    return BYZ_OK;
}

int launch_distances_from_gram(byz_ctx* ctx, const double* gram, int64_t n, float* dist, hipStream_t stream) {
    BYZ_REQUIRE(gram && dist && n > 0, "distances: bad arguments");
    KernelTimer t(ctx, BYZ_K_DISTANCES, stream);
    dim3 grid(static_cast<unsigned>(ceil_div(n, 64)), static_cast<unsigned>(ceil_div(n, 4)));
    distance_kernel<<<grid, 256, 0, stream>>>(gram, n, dist);
    return check_launch("distance_kernel");
}

}  // namespace byz
