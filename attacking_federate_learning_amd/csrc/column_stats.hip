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

#include <algorithm>
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
    float* __restrict__ mean_out, float* __restrict__ std_out, float* __restrict__ drift_out,
    const int32_t* __restrict__ redo_gate) {
    // redo_gate (optional): this launch stands behind the register-resident kernel and runs only if a wave of that kernel
    // gave up waiting for its turn (the word is then non-zero): the same bits, the slow way, instead of an invalid vector
    if (redo_gate != nullptr && __hip_atomic_load(redo_gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
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

// One LINK of the chain, for rows that are spread over several owners (the clients layout: the malicious clients' rows sit on the
// first ranks, sharded.py): out[c] = carry[c] (or +0.0) followed by this owner's rows in row order -- of x, or with `mean` of
// fl(fl(x - mean)^2).  The owners call it one after the other, each handing its `out` to the next as `carry`; the additions are
// then the very chain column_sequential_kernel runs over the stacked rows.
template <int VEC, bool SQUARES>
__global__ __launch_bounds__(kThreads) void column_chain_kernel(
    const float* __restrict__ G, int64_t n_rows, int64_t n_cols, int64_t ld, const float* __restrict__ carry,
    const float* __restrict__ mean, float* __restrict__ out) {
    const int64_t c0 = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * VEC;
    if (c0 >= n_cols) return;
    const bool full = c0 + VEC <= n_cols;
    float s[VEC], mu[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        const bool in = c0 + v < n_cols;
        s[v] = (carry != nullptr && in) ? carry[c0 + v] : 0.0f;
        mu[v] = (SQUARES && in) ? mean[c0 + v] : 0.0f;
    }
    const float* p = G + c0;
    int64_t r = 0;
    for (; r + kRowRun <= n_rows; r += kRowRun) {
        float x[kRowRun][VEC];
#pragma unroll
        for (int u = 0; u < kRowRun; ++u) load_columns<VEC>(p + (r + u) * ld, full, c0, n_cols, x[u]);
#pragma unroll
        for (int u = 0; u < kRowRun; ++u)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if constexpr (SQUARES) {
                    const float d = x[u][v] - mu[v];
                    const float q = d * d;
                    s[v] = s[v] + q;
                } else {
                    s[v] = s[v] + x[u][v];
                }
            }
    }
    for (; r < n_rows; ++r) {
        float x[VEC];
        load_columns<VEC>(p + r * ld, full, c0, n_cols, x);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if constexpr (SQUARES) {
                const float d = x[v] - mu[v];
                const float q = d * d;
                s[v] = s[v] + q;
            } else {
                s[v] = s[v] + x[v];
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v)
        if (c0 + v < n_cols) out[c0 + v] = s[v];
}

// The end of a chain: mean = sum / m; std = sqrt(sumsq / m), drift = mean - z std (either half optional).
__global__ __launch_bounds__(kThreads) void column_finish_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq,
                                                                 float rows_f, float num_std, int64_t n_cols,
                                                                 float* __restrict__ mean, float* __restrict__ stdev,
                                                                 float* __restrict__ drift) {
    const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (c >= n_cols) return;
    float m = 0.0f;
    if (sum != nullptr) {
        m = sum[c] / rows_f;
        mean[c] = m;
    } else if (sumsq != nullptr) {
        m = mean[c];
    }
    if (sumsq != nullptr) {
        const float sd = __builtin_sqrtf(sumsq[c] / rows_f);
        if (stdev != nullptr) stdev[c] = sd;
        if (drift != nullptr) drift[c] = m - num_std * sd;
    }
}

// ---- the attack's statistics with the rows RESIDENT IN REGISTERS between the two walks -----------------------------------
//
// The variance needs the mean first, so the rows are walked twice, and a matrix of m = 2400 rows x 3.125e6 columns (30 GB)
// does not come out of any cache the second time: column_sequential_kernel above moves 60 GB for 30 GB of input (10.3 ms,
// 0.37 of HBM; profiles/r05a_attack_sequential_*).  Here a workgroup keeps a tile of 32 columns x m rows in its REGISTERS
// (m = 2400: 300 KiB of the CU's 512 KiB) and both walks read registers; HBM is read once.
//
//   tile      32 columns (one 128-byte line per row) x all m rows.  NW waves; lanes 0-31 and 32-63 of a wave are two SEGMENTS
//             of consecutive rows (segment g = 2 wave + half holds rows g R .. g R + R - 1, R = ceil(m / (2 NW)) <= S = 8 RB); lane
//             (half, c) holds its segment's values of column c in x[0..R).  Rows past m hold +0.0.
//   chain     numpy adds in row order, so per column the additions are ONE chain through all segments: wave w takes its turn when
//             the workgroup's token says so, continues the running sum from `carry` (LDS), adds its lower segment (all lanes run
//             the adds; lanes 0-31 are the ones that count), hands the sum to lanes 32-63, adds the upper segment, leaves the
//             sum in `carry` and passes the token on.  Adding a padded +0.0 leaves a sum unchanged (the chain starts at +0.0 and
//             can never be -0.0), so every segment runs the same R additions.  The second walk is the same chain over
//             q = fl(fl(x - mean)^2), zero for padded rows.
//   pipeline  a wave that has finished its part of the second walk loads its rows of the NEXT tile while the token is with the
//             other waves: the loads hide behind the chain (2 x 2 NW R dependent additions per tile).
// The arithmetic is column_sequential_kernel's, operation for operation (the GPU tests hold both to numpy bit for bit).
constexpr int kTileCols = 32;

// The hand-off word of a column: (turn << 32) | bits of the running sum, ONE 8-byte LDS word written with one ds_write_b64 by
// the lane that finished a turn and polled with ds_read_b64 by the lane that takes the next -- the value arrives with its tag,
// no second read, no wait between a payload and a flag (a separate token word cost two dependent LDS round trips per hand-off:
// 7.7 ms instead of ... at m = 2400 x 3.125e6).  Volatile asm: the compiler may not move the poll across the other volatile
// statements, in particular not in front of register-only work that has to be finished before a wave starts waiting (an
// ordinary acquire load was hoisted above the squared deviations, which put them on the chain).
__device__ __forceinline__ unsigned lds_address(const void* p) {
    return static_cast<unsigned>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void*)p));
}
__device__ __forceinline__ unsigned long long lds_peek64(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_poke64(unsigned addr, unsigned long long v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// Bounded: a turn that never comes (it cannot, short of a bug) must not hang the device; the status word says so.
constexpr int kStatusNoTurn = 16;

template <int NW, int RB>
__global__ __launch_bounds__(NW * 64) void column_resident_kernel(
    const float* __restrict__ G, int n_rows, int64_t n_cols, int64_t ld, float num_std, float* __restrict__ mean_out,
    float* __restrict__ std_out, float* __restrict__ drift_out, int n_tiles, int32_t* __restrict__ status) {
    __shared__ __attribute__((aligned(8))) unsigned long long handoff[kTileCols];
    __shared__ float mean_s[kTileCols];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, c = lane & 31;
    constexpr int S = RB * 8;                              // register slots of a lane: the kernel is instantiated per RB, so
                                                           // that the chain is straight-line code (no test between additions)
    const int R = (n_rows + 2 * NW - 1) / (2 * NW);      // rows per segment, S - 8 < R <= S (the launcher's business)
    const int row0 = (2 * wave + half) * R;
    int mine = n_rows - row0;                              // rows of my segment
    mine = mine < 0 ? 0 : (mine > R ? R : mine);
    const float rows_f = static_cast<float>(n_rows);
    if (threadIdx.x < kTileCols) handoff[threadIdx.x] = 0ull;      // turn 0 may start
    __syncthreads();
    const unsigned my_word = lds_address(&handoff[c]);

    // tiles: XCD x (workgroups x, x + 8, ...) owns a contiguous range; its workgroups take neighbouring tiles at the same time,
    // so the 128-byte lines that a row shares between two tiles (rows are not line-aligned unless ld is a multiple of 32) are
    // asked for by the same L2 at about the same time
    // (a grid that is not a multiple of eight -- fewer tiles than CUs -- strides plainly)
    const bool grouped = (gridDim.x & 7) == 0;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per_xcd = gridDim.x >> 3;                    // workgroups of an XCD
    const int tiles_x = (n_tiles + 7) >> 3;
    auto tile_of = [&](int it) {
        if (!grouped) {
            const int64_t t = static_cast<int64_t>(it) * gridDim.x + blockIdx.x;
            return t < n_tiles ? static_cast<int>(t) : n_tiles;
        }
        const int local = it * per_xcd + slot;
        const int t = xcd * tiles_x + local;
        return local < tiles_x && t < n_tiles ? t : n_tiles;
    };

    float x[S];
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        const int64_t col = static_cast<int64_t>(tile) * kTileCols + c;
        const bool col_ok = col < n_cols;
        const float* p = G + static_cast<int64_t>(row0) * ld + col;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            x[s] = 0.0f;
            if (col_ok && s < mine) x[s] = p[0];
            p += ld;
        }
    };
    // one walk's share of the chain: S dependent additions, straight-line (the slots past R hold +0.0)
    auto add_segment = [&](float acc) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < S; ++s) acc = acc + x[s];
        return acc;
    };
    // wait until turn `turn` may start (the previous turn's owner has written its word); returns the value that came with it
    auto wait_turn = [&](int turn) __attribute__((always_inline)) -> float {
        unsigned spins = 0;
        for (;;) {
            const unsigned long long w = lds_peek64(my_word);
            const int tag = static_cast<int>(w >> 32);
            if (__all(tag >= turn)) return __uint_as_float(static_cast<uint32_t>(w));
            // the wave whose turn is next polls without a pause; the others can afford one
            if (turn - __builtin_amdgcn_readfirstlane(tag) > 1) __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 23)) {
                atomicOr(status, kStatusNoTurn);
                return 0.0f;
            }
        }
    };
    // my turn of a walk: continue the running sum through my two segments
    auto my_turn = [&](int turn) __attribute__((always_inline)) -> float {
        float acc = wait_turn(turn);
        if (wave == 0) acc = 0.0f;                         // a walk starts at add.reduce's identity
        acc = add_segment(acc);                            // lanes 0-31: the sum after the lower segment
        // ... handed to the lanes of the upper segment: lanes 32-63 take lanes 0-31's value (v_permlane32_swap_b32)
        acc = __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false)[0]);
        acc = add_segment(acc);                            // lanes 32-63: the sum after the upper segment
        return acc;
    };
    auto pass_on = [&](int turn, float value) __attribute__((always_inline)) {
        if (half == 1) lds_poke64(my_word, (static_cast<unsigned long long>(static_cast<uint32_t>(turn + 1)) << 32) | __float_as_uint(value));
    };

    int it = 0, base = 0;
    int tile = tile_of(0);
    if (tile < n_tiles) load_tile(tile);
    while (tile < n_tiles) {
        const int64_t col = static_cast<int64_t>(tile) * kTileCols + c;
        const bool writer = half == 1 && col < n_cols;     // the lanes that hold a finished chain
        // ---- first walk: the sum, the mean
        float acc = my_turn(base + wave);
        if (wave == NW - 1) {
            const float m = acc / rows_f;
            if (half == 1) mean_s[c] = m;                  // (LDS is in order per wave: in place before the word below)
            if (writer && mean_out) mean_out[col] = m;
        }
        pass_on(base + wave, acc);
        // ---- the squared deviations, in place (off the chain: every wave does its own as soon as the mean is known)
        (void)wait_turn(base + NW);
        const float mean = mean_s[c];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float d = x[s] - mean;
            const float q = d * d;
            x[s] = s < mine ? q : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) asm volatile("" : "+v"(x[s]));     // ... and they ARE done before the wait below
        // ---- second walk: the sum of squares, std, drift
        acc = my_turn(base + NW + wave);
        // the word goes on FIRST (issuing 8 RB loads takes thousands of cycles -- address arithmetic, the 63-deep request
        // counter -- and must not sit on the chain: with the loads in front of the hand-off a tile took 65 us instead of 14);
        // then, my registers being free, I fetch my rows of the next tile while the other waves finish this one
        pass_on(base + NW + wave, acc);
        if (wave == NW - 1 && writer) {
            const float var = acc / rows_f;
            const float sd = __builtin_sqrtf(var);
            if (std_out) std_out[col] = sd;
            if (drift_out) drift_out[col] = mean - num_std * sd;
        }
        ++it;
        tile = tile_of(it);
        if (tile < n_tiles) load_tile(tile);
        base += 2 * NW;
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
    // the attack over many rows: one pass over HBM with the tile resident in registers (BYZ_ATTACK_RESIDENT=0: the two-pass
    // kernel for everything -- the same bits, the A/B)
    constexpr int kMaxRb = 10;                   // 80 register slots per lane: 16 waves x 2 segments x 80 = 2560 rows
    if (stats && n_rows > 64 && n_rows <= 2 * 16 * 8 * kMaxRb && n_cols >= kTileCols && env_int("BYZ_ATTACK_RESIDENT", 1) != 0) {
        const int64_t n_tiles = ceil_div(n_cols, static_cast<int64_t>(kTileCols));
        BYZ_REQUIRE(n_tiles < (1 << 30), "column statistics: too many columns");
        // up to 640 rows four waves hold a tile, up to 1280 eight, up to 2560 sixteen (one workgroup per CU): the fewest waves
        // that hold the rows -- a hand-off between waves costs as much as thirty additions
        const int nw = n_rows <= 2 * 4 * 8 * kMaxRb ? 4 : n_rows <= 2 * 8 * 8 * kMaxRb ? 8 : 16;
        const int rb = static_cast<int>(ceil_div(ceil_div(n_rows, 2 * nw), 8));
        // (a lane's registers: 20 + 8 rb; a CU runs 32 waves at most)
        const int per_cu = (rb <= 5 ? 32 : 16) / nw;
        const int64_t wgs = std::min<int64_t>(n_tiles, static_cast<int64_t>(ctx->num_cus) * per_cu);
#define BYZ_RESIDENT(NW, RB)                                                                                         \
    case RB:                                                                                                         \
        column_resident_kernel<NW, RB><<<static_cast<unsigned>(wgs), NW * 64, 0, stream>>>(                          \
            G, static_cast<int>(n_rows), n_cols, ld, num_std, mean, stdev, drift, static_cast<int>(n_tiles),         \
            attack_redo_word(ctx));                                                                                  \
        break
#define BYZ_RESIDENT_ALL(NW)                                                                                         \
    switch (rb) {                                                                                                    \
        BYZ_RESIDENT(NW, 1); BYZ_RESIDENT(NW, 2); BYZ_RESIDENT(NW, 3); BYZ_RESIDENT(NW, 4); BYZ_RESIDENT(NW, 5);       \
        BYZ_RESIDENT(NW, 6); BYZ_RESIDENT(NW, 7); BYZ_RESIDENT(NW, 8); BYZ_RESIDENT(NW, 9); BYZ_RESIDENT(NW, 10);      \
        default: set_error("column statistics: %d row blocks per segment", rb); return BYZ_E_INVALID;                \
    }
        // a wave that never gets its turn (a bounded spin: it has not been seen to happen) sets the redo word and carries on
        // with +0.0; the two-pass kernel behind it then recomputes every column -- nothing of an invalid vector survives the
        // call, and nothing is reported by a later, unrelated one
        // (BYZ_ATTACK_FORCE_REDO=1: the word starts non-zero -- the test of the path that has not been seen to happen)
        BYZ_HIP(hipMemsetAsync(attack_redo_word(ctx), env_int("BYZ_ATTACK_FORCE_REDO", 0) != 0 ? 1 : 0, sizeof(int32_t), stream));
        if (nw == 4) { BYZ_RESIDENT_ALL(4) } else if (nw == 8) { BYZ_RESIDENT_ALL(8) } else { BYZ_RESIDENT_ALL(16) }
#undef BYZ_RESIDENT_ALL
#undef BYZ_RESIDENT
        BYZ_TRY(check_launch("column_resident_kernel"));
        const dim3 redo_grid(static_cast<unsigned>(col_blocks));
        if (vec4) column_sequential_kernel<4, 1><<<redo_grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, num_std, mean, stdev, drift, attack_redo_word(ctx));
        else column_sequential_kernel<1, 1><<<redo_grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, num_std, mean, stdev, drift, attack_redo_word(ctx));
        return check_launch("column_sequential_kernel (redo)");
    }
    const dim3 grid(static_cast<unsigned>(col_blocks));
    if (stats) {
        if (vec4) column_sequential_kernel<4, 1><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, num_std, mean, stdev, drift, nullptr);
        else column_sequential_kernel<1, 1><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, num_std, mean, stdev, drift, nullptr);
    } else {
        if (vec4) column_sequential_kernel<4, 0><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, 0.0f, mean, nullptr, nullptr, nullptr);
        else column_sequential_kernel<1, 0><<<grid, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, 0.0f, mean, nullptr, nullptr, nullptr);
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

int launch_column_chain(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const float* carry,
                        const float* mean, float* out, hipStream_t stream) {
    BYZ_REQUIRE(G && out && n_rows > 0 && n_cols > 0 && ld >= n_cols, "column chain: bad shape %lld x %lld ld %lld",
                (long long)n_rows, (long long)n_cols, (long long)ld);
    const bool vec4 = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0) &&
                      n_cols >= static_cast<int64_t>(4) * kThreads * ctx->num_cus * 2;
    const unsigned blocks = static_cast<unsigned>(ceil_div(n_cols, static_cast<int64_t>(kThreads) * (vec4 ? 4 : 1)));
    KernelTimer t(ctx, BYZ_K_COLUMN_STATS, stream);
    if (mean != nullptr) {
        if (vec4) column_chain_kernel<4, true><<<blocks, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, carry, mean, out);
        else column_chain_kernel<1, true><<<blocks, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, carry, mean, out);
    } else {
        if (vec4) column_chain_kernel<4, false><<<blocks, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, carry, nullptr, out);
        else column_chain_kernel<1, false><<<blocks, kThreads, 0, stream>>>(G, n_rows, n_cols, ld, carry, nullptr, out);
    }
    return check_launch("column_chain_kernel");
}

int launch_column_finish(byz_ctx* ctx, const float* sum, const float* sumsq, int64_t total_rows, float num_std, int64_t n_cols,
                         float* mean, float* stdev, float* drift, hipStream_t stream) {
    BYZ_REQUIRE(sum || sumsq, "column finish: neither a sum nor a sum of squared deviations");
    BYZ_REQUIRE(mean, "column finish: `mean` is the output of the sum's half and the input of the squared deviations' half");
    BYZ_REQUIRE(total_rows > 0 && n_cols > 0, "column finish: bad shape %lld rows, %lld columns", (long long)total_rows, (long long)n_cols);
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    column_finish_kernel<<<static_cast<unsigned>(ceil_div(n_cols, kThreads)), kThreads, 0, stream>>>(
        sum, sumsq, static_cast<float>(total_rows), num_std, n_cols, mean, stdev, drift);
    return check_launch("column_finish_kernel");
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
