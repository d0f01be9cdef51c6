// Column statistics over the rows of the gradient matrix: HBM-bound streaming kernels.
//
//   no_defense            reference defences.py:13-14   out[c] = mean_r G[r][c]
//   Attack.attack         reference malicious.py:18-19  mean / population std over the malicious rows
//   DriftAttack hook      reference malicious.py:34-36  mean - z * std
//   Server.defend update  reference server.py:89-90     v = mu*v - lr*agg ; w += v
//
// Layout: G is row-major, so a wave reading 64 (or 256, with dwordx4) consecutive columns of one row is
// a single coalesced request; each thread walks down the rows of its own column(s) and adds them in row order in fp32:
// the reference's numpy arithmetic operation by operation, so mean, std and the drifted vector are the reference's BITS
// (the fp64 one-pass statistics of rounds 1-4 were 1 ulp off in a few columns, which moved median-window
// decisions downstream).  The variance needs the mean first: the rows are walked twice.
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

// numpy's arithmetic for np.mean(rows, axis=0) and np.var(rows, axis=0) ** 0.5 on an (m, D) fp32 array, operation by operation
// (tests/test_oracle_golden.py pins this model to numpy itself):
//     s    = x[0] + x[1] + ... + x[m-1]          sequential fp32, in row order (a reduction over the OUTER axis is not pairwise)
//     mean = s / float(m)
//     s2   = sum over rows, sequential fp32, of  fl(fl(x[r] - mean) * fl(x[r] - mean))
//     std  = sqrt(s2 / float(m))                 (`** 0.5` on an fp32 array is np.sqrt)
//     drift = mean - fl(float(z) * std)
// One thread owns VEC columns and walks the rows; the loads of a run of rows are issued together, the additions follow in row
// order.  The chain starts from +0.0f, add.reduce's identity (a column of -0.0 sums to +0.0 in numpy, and here).
// This file is compiled with -ffp-contract=off: no multiply may fuse with the addition that follows it.
constexpr int kRowRun = 8;

template <int VEC>
__device__ __forceinline__ void load_columns(const float* __restrict__ p, bool full, int64_t c0, int64_t n_cols, float (&x)[VEC]) {
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
}

// MODE 0: mean only (no_defense).  MODE 1: mean, std, drift (the attack).
// carry_in (optional, [VEC columns]): the chain continues a sum begun over earlier rows that live elsewhere (another rank's
// clients); `total_rows` is the divisor (all rows of the chain, not only the local ones).
template <int VEC, int MODE>
__global__ __launch_bounds__(kThreads) void column_sequential_kernel(
    const float* __restrict__ G, int64_t n_rows, int64_t n_cols, int64_t ld, float num_std,
    float* __restrict__ mean_out, float* __restrict__ std_out, float* __restrict__ drift_out) {
    const int64_t c0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * VEC;
    if (c0 >= n_cols) return;
    const bool full = c0 + VEC <= n_cols;
    const float rows_f = static_cast<float>(n_rows);
    float s[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = 0.0f;
    const float* p = G + c0;
    int64_t r = 0;
    for (; r + kRowRun <= n_rows; r += kRowRun) {
        float x[kRowRun][VEC];
#pragma unroll
        for (int u = 0; u < kRowRun; ++u) load_columns<VEC>(p + (r + u) * ld, full, c0, n_cols, x[u]);
#pragma unroll
        for (int u = 0; u < kRowRun; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[v] = s[v] + x[u][v];
    }
    for (; r < n_rows; ++r) {
        float x[VEC];
        load_columns<VEC>(p + r * ld, full, c0, n_cols, x);
#pragma unroll
        for (int v = 0; v < VEC; ++v) s[v] = s[v] + x[v];
    }
    float mean[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        mean[v] = s[v] / rows_f;
        if (mean_out && c0 + v < n_cols) mean_out[c0 + v] = mean[v];
    }
    if constexpr (MODE == 1) {
        float s2[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) s2[v] = 0.0f;
        r = 0;
        for (; r + kRowRun <= n_rows; r += kRowRun) {
            float x[kRowRun][VEC];
#pragma unroll
            for (int u = 0; u < kRowRun; ++u) load_columns<VEC>(p + (r + u) * ld, full, c0, n_cols, x[u]);
#pragma unroll
            for (int u = 0; u < kRowRun; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float d = x[u][v] - mean[v];
                    const float q = d * d;
                    s2[v] = s2[v] + q;
                }
        }
        for (; r < n_rows; ++r) {
            float x[VEC];
            load_columns<VEC>(p + r * ld, full, c0, n_cols, x);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float d = x[v] - mean[v];
                const float q = d * d;
                s2[v] = s2[v] + q;
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if (c0 + v >= n_cols) continue;
            const float var = s2[v] / rows_f;
            const float sd = __builtin_sqrtf(var);          // correctly rounded (hipcc's default for fp32 sqrt and divide)
            if (std_out) std_out[c0 + v] = sd;
            if (drift_out) drift_out[c0 + v] = mean[v] - num_std * sd;     // malicious.py:35 on the fp32 values
        }
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
    // 16-byte loads when every row starts 16-byte aligned and the columns alone fill the chip; one column per thread otherwise
    // (few columns: four times the threads)
    const bool vec4 = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0) &&
                      n_cols >= static_cast<int64_t>(4) * kThreads * ctx->num_cus * 2;
    const int64_t col_blocks = ceil_div(n_cols, static_cast<int64_t>(kThreads) * (vec4 ? 4 : 1));
    KernelTimer t(ctx, BYZ_K_COLUMN_STATS, stream);
    const dim3 grid(static_cast<unsigned>(col_blocks));
    if (stats) {
        if (vec4) column_sequential_kernel<4, 1><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, num_std, mean, stdev, drift);
        else column_sequential_kernel<1, 1><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, num_std, mean, stdev, drift);
    } else {
        if (vec4) column_sequential_kernel<4, 0><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, 0.0f, mean, nullptr, nullptr);
        else column_sequential_kernel<1, 0><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, 0.0f, mean, nullptr, nullptr);
    }
    return check_launch("column_sequential_kernel");
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
    // 4 KiB pieces of a row a workgroup writes back to back: one of the instantiated 1, 2, 4, 8, 16 (anything else is 16 --
    // the grid below must be sized for the RUN that is launched)
    int run = env_int("BYZ_BROADCAST_RUN", 16);
    if (run != 1 && run != 2 && run != 4 && run != 8) run = 16;
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
