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

// 4 consecutive floats starting at column k, zero-filled from k_hi on
__device__ __forceinline__ f32x4 load_tail(const float* __restrict__ p, int64_t k, int64_t k_hi) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k + 0 < k_hi) v.x = p[0];
    if (k + 1 < k_hi) v.y = p[1];
    if (k + 2 < k_hi) v.z = p[2];
    if (k + 3 < k_hi) v.w = p[3];
    return v;
}

template <typename PartialT>
__global__ __launch_bounds__(THREADS, 2) void gram_tile_kernel(const float* __restrict__ G, int64_t n_rows,
                                                               int64_t n_cols, int64_t ld,
                                                               int64_t stages_per_split,
                                                               PartialT* __restrict__ partial, int n_tiles,
                                                               const int2* __restrict__ tile_order, int per_xcd, int n_splits) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_FLOATS];  // [2 buffers][A | B][TM][LDS_STRIDE]

    // Workgroup -> (tile, K split).  Workgroups are dealt to the 8 XCDs round-robin (observed; only speed
    // depends on it), each XCD with its own L2.  With many tiles, XCD x works through its own contiguous
    // share of the tile list, split by split, and the list is ordered in 8 x 8 super-blocks of the tile
    // triangle: the ~64 workgroups an XCD runs at a time then share 8 + 8 operand row blocks instead of
    // streaming 128.
    int split, t_list;
    if (per_xcd > 0) {
        const int xcd = blockIdx.x & 7;
        const int seq = blockIdx.x >> 3;
        // XCD x owns tiles [first, first + mine): shares differ by at most one tile
        const int base = n_tiles >> 3, rem = n_tiles & 7;
        const int mine = base + (xcd < rem ? 1 : 0);
        const int first = xcd * base + (xcd < rem ? xcd : rem);
        split = seq / mine;
        if (split >= n_splits) return;
        t_list = first + (seq - split * mine);
    } else {  // few tiles: plain tile-fastest order, every XCD busy
        split = blockIdx.x / n_tiles;
        t_list = blockIdx.x - split * n_tiles;
    }
    const int2 tt = tile_order[t_list];
    const int ti = tt.x, tj = tt.y;
    const int tile = ti * (ti + 1) / 2 + tj;
    const bool diagonal = (ti == tj);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    const int64_t k_begin = static_cast<int64_t>(split) * stages_per_split * BK;
    int64_t k_end = k_begin + stages_per_split * BK;
    if (k_end > n_cols) k_end = n_cols;
    const int n_stages = k_begin < k_end ? static_cast<int>((k_end - k_begin + BK - 1) / BK) : 0;
    const int n_full = k_begin < k_end ? static_cast<int>((k_end - k_begin) / BK) : 0;   // stages without a ragged K tail

    // staging assignment: 8 lanes cover one 128-byte row segment, 32 rows per pass, 4 passes per operand.
    // Rows past the matrix are clamped to the last row: the duplicates land in Gram entries nobody reads,
    // and the main loop carries no per-load branch (a branch around a load makes hipcc drain vmcnt).
    const int ld_kq = (tid & 7) * 4;
    const int ld_row = tid >> 3;
    const float* a_ptr[4];
    const float* b_ptr[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int64_t ra_ = static_cast<int64_t>(ti) * TM + ld_row + 32 * p;
        int64_t rb_ = static_cast<int64_t>(tj) * TM + ld_row + 32 * p;
        if (ra_ > n_rows - 1) ra_ = n_rows - 1;
        if (rb_ > n_rows - 1) rb_ = n_rows - 1;
        a_ptr[p] = G + ra_ * ld + ld_kq;
        b_ptr[p] = G + rb_ * ld + ld_kq;
    }

    f32x4 ra[4], rb[4];
    // the main loop only ever issues unconditional 16-byte loads; the ragged K tail has its own code
    auto fetch_full = [&](int stage) {
        const int64_t k = k_begin + static_cast<int64_t>(stage) * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const f32x4u*>(a_ptr[p] + k);
        if (!diagonal) {
#pragma unroll
            for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const f32x4u*>(b_ptr[p] + k);
        }
    };
    auto fetch_any = [&](int stage) {
        if (stage < n_full) {
            fetch_full(stage);
        } else {  // the one ragged stage of the last split: zero-fill past k_end
            const int64_t k = k_begin + static_cast<int64_t>(stage) * BK;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                ra[p] = load_tail(a_ptr[p] + k, k + ld_kq, k_end);
                if (!diagonal) rb[p] = load_tail(b_ptr[p] + k, k + ld_kq, k_end);
            }
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

    auto compute = [&](int s) {
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
    };
    auto flush = [&](int s) {
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
    };

    if (n_stages > 0) {
        fetch_any(0);
        stash(0);
    }
    __syncthreads();
    int s = 0;
    for (; s + 1 < n_full; ++s) {   // steady state: the next stage is a full one
        fetch_full(s + 1);
        compute(s);
        stash((s + 1) & 1);
        __syncthreads();
        flush(s);
    }
    for (; s < n_stages; ++s) {     // at most two stages: the last full one and the ragged tail
        if (s + 1 < n_stages) fetch_any(s + 1);
        compute(s);
        if (s + 1 < n_stages) stash((s + 1) & 1);
        __syncthreads();
        flush(s);
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
    // split-K.  Two workgroups fit a CU (LDS), so the chip runs `slots` workgroups at a time; the grid is
    // n_tiles * splits of them, all of equal length.  Pick the split count whose last round of workgroups
    // is (nearly) full -- 528 tiles x 2 splits would leave the chip one third idle, 528 x 31 does not --
    // while keeping at least `min_stages` K stages (512 columns) per slab so that slab traffic stays a small
    // fraction of the matrix traffic.
    const int64_t slots = static_cast<int64_t>(ctx->num_cus) * 2;
    const int64_t min_stages = env_int("BYZ_GRAM_MIN_STAGES", 16);
    int64_t max_splits = stages / min_stages;
    if (max_splits < 1) max_splits = 1;
    if (max_splits > 4096) max_splits = 4096;
    int64_t splits = 1;
    {
        double best = -1.0;
        const int64_t want = ceil_div(slots * 3, n_tiles);   // at least ~3 rounds when K allows it
        for (int64_t s = 1; s <= max_splits; ++s) {
            const int64_t wgs = n_tiles * s;
            const double eff = static_cast<double>(wgs) / static_cast<double>(ceil_div(wgs, slots) * slots);
            // prefer fuller last rounds; among equals the fewer slabs; below `want` only if nothing else fits
            const double score = eff - (s < want ? 0.05 : 0.0) - 1e-4 * static_cast<double>(s > want ? s - want : 0);
            if (score > best) {
                best = score;
                splits = s;
            }
        }
    }
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
    // tile list in 8 x 8 super-block order (see the kernel); rebuilt only when the tile count changes
    if (ctx->tile_order_T != T) {
        ctx->tile_order_host.clear();
        const int64_t S = ceil_div(T, 8);
        for (int64_t I = 0; I < S; ++I)
            for (int64_t J = 0; J <= I; ++J)
                for (int64_t ti = I * 8; ti < I * 8 + 8 && ti < T; ++ti)
                    for (int64_t tj = J * 8; tj < J * 8 + 8 && tj <= ti; ++tj) {
                        ctx->tile_order_host.push_back(static_cast<int32_t>(ti));
                        ctx->tile_order_host.push_back(static_cast<int32_t>(tj));
                    }
        BYZ_TRY(ctx->tile_order.ensure(ctx->tile_order_host.size() * sizeof(int32_t)));
        BYZ_HIP(hipMemcpyAsync(ctx->tile_order.ptr, ctx->tile_order_host.data(),
                               ctx->tile_order_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BYZ_HIP(hipStreamSynchronize(stream));   // the host vector is pageable; once per matrix height
        ctx->tile_order_T = T;
    }
    // XCD-partitioned order only when every XCD gets enough tiles for its shares to be even (<= 3% apart)
    const bool partitioned = n_tiles >= 256;
    const int64_t per_xcd = partitioned ? ceil_div(n_tiles, 8) : 0;
    const int64_t grid_wgs = partitioned ? 8 * per_xcd * splits : n_tiles * splits;
    if (grid_wgs > 0x7fffffff) {
        set_error("gram: grid too large");
        return BYZ_E_UNSUPPORTED;
    }
    {
        KernelTimer t(ctx, BYZ_K_GRAM, stream);
        const int2* order = ctx->tile_order.as<int2>();
        if (wide)
            gram_tile_kernel<double><<<static_cast<unsigned>(grid_wgs), THREADS, 0, stream>>>(
                G, n_rows, n_cols, ld, stages_per_split, ctx->gram_partials.as<double>(), (int)n_tiles, order, (int)per_xcd, (int)splits);
        else
            gram_tile_kernel<float><<<static_cast<unsigned>(grid_wgs), THREADS, 0, stream>>>(
                G, n_rows, n_cols, ld, stages_per_split, ctx->gram_partials.as<float>(), (int)n_tiles, order, (int)per_xcd, (int)splits);
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
