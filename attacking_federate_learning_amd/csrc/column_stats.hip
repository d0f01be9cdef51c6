// Column statistics over the rows of the gradient matrix: HBM-bound streaming kernels.
//
//   no_defense            reference defences.py:13-14   out[c] = mean_r G[r][c]
//   Attack.attack         reference malicious.py:18-19  mean / population std over the malicious rows
//   DriftAttack hook      reference malicious.py:34-36  mean - z * std
//   Server.defend update  reference server.py:89-90     v = mu*v - lr*agg ; w += v
//
// Layout: G is row-major, so a wave reading 64 (or 256, with dwordx4) consecutive columns of one row is
// a single coalesced request; each thread walks down the rows of its own column(s).  Sums are carried in
// fp64 (free: the kernel moves 4 bytes per fp64 add) and the variance is taken about the first row's
// value, so the fp32 reference result (two-pass) is reproduced to ~1 ulp without a second pass over HBM.
// Algorithmic traffic: 4*rows*cols bytes read + 4*cols written.
#include "common.hpp"

#include <cstdlib>

namespace byz {
namespace {

int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : fallback;
}


constexpr int kThreads = 256;
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

// partial layout: [split][2][n_cols] doubles (plane 0: sum of (x - x0), plane 1: sum of squares)
template <int VEC, bool STATS>
__global__ __launch_bounds__(kThreads) void column_partial_kernel(
    const float* __restrict__ G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t rows_per_split,
    double* __restrict__ partial) {
    const int64_t c0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * VEC;
    if (c0 >= n_cols) return;
    const int64_t r_begin = static_cast<int64_t>(blockIdx.y) * rows_per_split;
    const int64_t r_end = r_begin + rows_per_split < n_rows ? r_begin + rows_per_split : n_rows;
    double s1[VEC], s2[VEC];
    float x0[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        s1[v] = 0.0;
        s2[v] = 0.0;
        x0[v] = (STATS && c0 + v < n_cols) ? G[c0 + v] : 0.0f;
    }
    const float* p = G + r_begin * ld + c0;
    const bool full = c0 + VEC <= n_cols;
#pragma unroll 4
    for (int64_t r = r_begin; r < r_end; ++r, p += ld) {
        float x[VEC];
        if constexpr (VEC == 4) {
            if (full) {
                const float4u q = *reinterpret_cast<const float4u*>(p);
                x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) x[v] = (c0 + v < n_cols) ? p[v] : 0.0f;
            }
        } else {
            x[0] = p[0];
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const double d = static_cast<double>(x[v]) - static_cast<double>(x0[v]);
            s1[v] += d;
            if (STATS) s2[v] += d * d;
        }
    }
    double* out = partial + static_cast<int64_t>(blockIdx.y) * 2 * n_cols;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        if (c0 + v < n_cols) {
            out[c0 + v] = s1[v];
            if (STATS) out[n_cols + c0 + v] = s2[v];
        }
    }
}

// Combines the row-split partials in a fixed order and writes mean / std / drift.
template <bool STATS>
__global__ __launch_bounds__(kThreads) void column_finalize_kernel(
    const double* __restrict__ partial, const float* __restrict__ G, int64_t n_rows, int64_t n_cols,
    int splits, float num_std, float* __restrict__ mean_out, float* __restrict__ std_out,
    float* __restrict__ drift_out) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (c >= n_cols) return;
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < splits; ++s) {
        s1 += partial[static_cast<int64_t>(s) * 2 * n_cols + c];
        if (STATS) s2 += partial[static_cast<int64_t>(s) * 2 * n_cols + n_cols + c];
    }
    const double inv = 1.0 / static_cast<double>(n_rows);
    const double shift = STATS ? static_cast<double>(G[c]) : 0.0;
    const double m1 = s1 * inv;
    const float mean = static_cast<float>(shift + m1);
    if (mean_out) mean_out[c] = mean;
    if (STATS) {
        double var = s2 * inv - m1 * m1;
        var = var > 0.0 ? var : 0.0;
        const float sd = static_cast<float>(sqrt(var));
        if (std_out) std_out[c] = sd;
        // malicious.py:35 evaluates mean - z*std on the fp32 values
        if (drift_out) drift_out[c] = __fsub_rn(mean, __fmul_rn(num_std, sd));
    }
}

// vec -> every row of G (malicious.py:26-27: all malicious clients get ONE array).  VEC = 4: 16-byte stores (rows and the
// vector 16-byte aligned), a workgroup owns RUN x 4 KiB CONSECUTIVE bytes of every row it visits -- with one 4 KiB piece per
// row visit (round 3) every wave's next store lay a whole row further on and HBM saw 1 KiB writes scattered over thousands of
// rows: 8.0 ms for 2400 rows x 3.125e6 columns (30 GB), 6.9 / 6.1 / 5.9 ms with RUN = 4 / 8 / 16 (5.1 TB/s;
// scripts/broadcast_probe.py, BYZ_BROADCAST_RUN); VEC = 1: any alignment.
template <int VEC, int RUN>
__global__ __launch_bounds__(kThreads) void broadcast_rows_kernel(float* __restrict__ G, int64_t n_rows,
                                                                  int64_t n_cols, int64_t ld,
                                                                  const float* __restrict__ vec) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int64_t c0 = (static_cast<int64_t>(blockIdx.x) * RUN * kThreads + threadIdx.x) * VEC;
    if constexpr (VEC == 4) {
        f32x4 v[RUN];
        bool whole[RUN];
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
            const int64_t c = c0 + static_cast<int64_t>(k) * kThreads * VEC;
            whole[k] = c + 3 < n_cols;
            v[k] = whole[k] ? *reinterpret_cast<const f32x4*>(vec + c) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        for (int64_t r = blockIdx.y; r < n_rows; r += gridDim.y) {
            float* row = G + r * ld;
#pragma unroll
            for (int k = 0; k < RUN; ++k) {
                const int64_t c = c0 + static_cast<int64_t>(k) * kThreads * VEC;
                if (whole[k]) *reinterpret_cast<f32x4*>(row + c) = v[k];
            }
        }
        // the ragged tail of the row (fewer than four columns): one thread's business
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
            const int64_t c = c0 + static_cast<int64_t>(k) * kThreads * VEC;
            if (!whole[k] && c < n_cols)
                for (int64_t e = c; e < n_cols; ++e) {
                    const float x = vec[e];
                    for (int64_t r = blockIdx.y; r < n_rows; r += gridDim.y) G[r * ld + e] = x;
                }
        }
    } else {
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
            const int64_t c = c0 + static_cast<int64_t>(k) * kThreads;
            if (c >= n_cols) continue;
            const float x = vec[c];
            for (int64_t r = blockIdx.y; r < n_rows; r += gridDim.y) G[r * ld + c] = x;
        }
    }
}

__global__ __launch_bounds__(kThreads) void drift_axpy_kernel(float* __restrict__ mean,
                                                              const float* __restrict__ stdev, int64_t n,
                                                              float num_std) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) mean[i] = __fsub_rn(mean[i], __fmul_rn(num_std, stdev[i]));
}

__global__ __launch_bounds__(kThreads) void server_update_kernel(float* __restrict__ w, float* __restrict__ v,
                                                                 const float* __restrict__ agg, int64_t n,
                                                                 float momentum, float lr) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (i < n) {
        // server.py:89-90, same operation order in fp32: (mu*v) - (lr*agg), then w + v
        const float nv = __fsub_rn(__fmul_rn(momentum, v[i]), __fmul_rn(lr, agg[i]));
        v[i] = nv;
        w[i] = __fadd_rn(w[i], nv);
    }
}

__global__ __launch_bounds__(kThreads) void copy_row_kernel(const float* __restrict__ G, int64_t ld,
                                                            int64_t n_rows, int64_t n_cols,
                                                            const int32_t* __restrict__ index,
                                                            float* __restrict__ out) {
    int64_t r = *index;
    if (r < 0) r += n_rows;  // numpy's G[-1]: the reference returns the last row when nothing won
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (c < n_cols) out[c] = G[r * ld + c];
}

int column_pass(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, bool stats,
                float num_std, float* mean, float* stdev, float* drift, hipStream_t stream) {
    BYZ_REQUIRE(G && n_rows > 0 && n_cols > 0 && ld >= n_cols, "column statistics: bad shape %lld x %lld ld %lld",
                (long long)n_rows, (long long)n_cols, (long long)ld);
    const bool vec4 = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0) && n_cols >= 4 * kThreads;
    const int vec = vec4 ? 4 : 1;
    const int64_t col_blocks = ceil_div(n_cols, static_cast<int64_t>(kThreads) * vec);
    // enough workgroups to cover the chip a few times over; rows are split when columns alone cannot
    int64_t splits = ceil_div(static_cast<int64_t>(ctx->num_cus) * 8, col_blocks);
    if (splits > ceil_div(n_rows, 8)) splits = ceil_div(n_rows, 8);
    if (splits < 1) splits = 1;
    const int64_t rows_per_split = ceil_div(n_rows, splits);
    splits = ceil_div(n_rows, rows_per_split);
    BYZ_TRY(ctx->colstat_partials.ensure(static_cast<size_t>(splits) * 2 * n_cols * sizeof(double)));
    double* partial = ctx->colstat_partials.as<double>();
    {
        KernelTimer t(ctx, BYZ_K_COLUMN_STATS, stream);
        dim3 grid(static_cast<unsigned>(col_blocks), static_cast<unsigned>(splits));
        if (stats) {
            if (vec4) column_partial_kernel<4, true><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, rows_per_split, partial);
            else column_partial_kernel<1, true><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, rows_per_split, partial);
        } else {
            if (vec4) column_partial_kernel<4, false><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, rows_per_split, partial);
            else column_partial_kernel<1, false><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, rows_per_split, partial);
        }
        BYZ_TRY(check_launch("column_partial_kernel"));
    }
    {
        KernelTimer t(ctx, BYZ_K_MISC, stream);
        const unsigned blocks = static_cast<unsigned>(ceil_div(n_cols, kThreads));
        if (stats)
            column_finalize_kernel<true><<<blocks, kThreads, 0, stream>>>(partial, G, n_rows, n_cols, (int)splits, num_std, mean, stdev, drift);
        else
            column_finalize_kernel<false><<<blocks, kThreads, 0, stream>>>(partial, G, n_rows, n_cols, (int)splits, 0.0f, mean, nullptr, nullptr);
        BYZ_TRY(check_launch("column_finalize_kernel"));
    }
    (void)vec;
    return BYZ_OK;
}

}  // namespace

int launch_column_mean(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* out,
                       hipStream_t stream) {
    BYZ_REQUIRE(out, "no_defense: null output");
    return column_pass(ctx, G, n_rows, n_cols, ld, false, 0.0f, out, nullptr, nullptr, stream);
}

int launch_column_drift(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float num_std,
                        float* drift, float* mean, float* stdev, hipStream_t stream) {
    return column_pass(ctx, G, n_rows, n_cols, ld, true, num_std, mean, stdev, drift, stream);
}

int launch_broadcast_rows(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const float* vec,
                          hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    const bool wide = ld % 4 == 0 && (reinterpret_cast<uintptr_t>(G) & 15u) == 0 && (reinterpret_cast<uintptr_t>(vec) & 15u) == 0;
    const int run = env_int("BYZ_BROADCAST_RUN", 16);      // 4 KiB pieces of a row a workgroup writes back to back (1, 2, 4, 8, 16)
    const int64_t per_wg = static_cast<int64_t>(kThreads) * (wide ? 4 : 1) * run;
    const unsigned blocks = static_cast<unsigned>(ceil_div(n_cols, per_wg));
    unsigned ysplit = static_cast<unsigned>(ceil_div(static_cast<int64_t>(ctx->num_cus) * 8, blocks));
    if (ysplit > n_rows) ysplit = static_cast<unsigned>(n_rows);
    if (ysplit < 1) ysplit = 1;
    const dim3 grid(blocks, ysplit);
#define BYZ_BC(V, R) broadcast_rows_kernel<V, R><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, vec)
    if (wide) {
        if (run == 1) BYZ_BC(4, 1);
        else if (run == 2) BYZ_BC(4, 2);
        else if (run == 4) BYZ_BC(4, 4);
        else if (run == 8) BYZ_BC(4, 8);
        else BYZ_BC(4, 16);
    } else {
        if (run == 1) BYZ_BC(1, 1);
        else if (run == 2) BYZ_BC(1, 2);
        else if (run == 4) BYZ_BC(1, 4);
        else if (run == 8) BYZ_BC(1, 8);
        else BYZ_BC(1, 16);
    }
#undef BYZ_BC
    return check_launch("broadcast_rows_kernel");
}

int launch_drift_axpy(byz_ctx* ctx, float* mean, const float* stdev, int64_t n, float num_std,
                      hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    drift_axpy_kernel<<<static_cast<unsigned>(ceil_div(n, kThreads)), kThreads, 0, stream>>>(mean, stdev, n, num_std);
    return check_launch("drift_axpy_kernel");
}

int launch_server_update(byz_ctx* ctx, float* w, float* v, const float* agg, int64_t n, float momentum, float lr,
                         hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    server_update_kernel<<<static_cast<unsigned>(ceil_div(n, kThreads)), kThreads, 0, stream>>>(w, v, agg, n, momentum, lr);
    return check_launch("server_update_kernel");
}

int launch_copy_row(byz_ctx* ctx, const float* G, int64_t ld, int64_t n_rows, int64_t n_cols,
                    const int32_t* index_dev, float* out, hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    copy_row_kernel<<<static_cast<unsigned>(ceil_div(n_cols, kThreads)), kThreads, 0, stream>>>(G, ld, n_rows, n_cols, index_dev, out);
    return check_launch("copy_row_kernel");
}

}  // namespace byz
