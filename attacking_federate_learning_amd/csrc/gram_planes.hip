// The Gram contraction of defences.py:16-21 on PRE-SPLIT operands (the long-K, many-tile case: N >= 2817, D > 16384).
//
// gram.hip's split arithmetic turns every fp32 operand into 16-bit planes inside the tile kernel, i.e. once per (tile,
// element): at N = 4000 every element of G is split ~32 times, and that VALU work (and its power: the kernel runs at the
// socket's power cap) is what holds the matrix pipe at a third of its rate.  Here the split runs ONCE per element:
//
//   plane_split_*_kernel   G[:, k0 : k0 + SC] (fp32, row-major)  ->  16-bit planes of that column range, stored in MFMA
//                          FRAGMENT ORDER: for every (32-row block, 16-column step, plane) the 1 KiB image that the 64
//                          lanes of a wave hold as the A/B operand of v_mfma_f32_32x32x16_{bf16,f16} (lane l: row l & 31,
//                          columns 8 (l >> 5) .. + 7).  HBM bound.
//   gram_planes_kernel     one workgroup = 8 waves = a 256 x 128 tile of C = G G^T (two 128 x 128 slabs of gram.hip's
//                          slab format), each wave a 64 x 64 sub-tile.  A 16-column stage of the 12 row blocks reaches LDS
//                          by LDS-DMA as contiguous 1 KiB pieces (lane-linear on both sides: perfectly coalesced,
//                          conflict-free ds_read_b128, no swizzle), NBUF stages of LDS, the DMA NBUF stages ahead, and the
//                          MFMAs of stage s run from registers while the fragments of stage s + 1 are read.
//
// Two arithmetics (BYZ_GRAM_MODE):
//   f16x2  (default here)  per (row, 8192-column chunk) the values are scaled by a power of two so that the chunk's
//          largest magnitude lies in [2^14, 2^15), then x = h + m + r with h = fp16(x), m = fp16(x - h) (both round to
//          nearest), |r| <= 2^-23 |x|: the operand's own fp32 rounding unit.  Three fp16 MFMAs per block form
//          m h' + h m' + h h' (each product exact in fp32); the dropped m m' is below 2^-22 |x y| with a random sign, except
//          on the diagonal where sum m^2 = ~4e-8 sum x^2.  Against fp64: 6e-8 relative on c_ii, 1e-9 on c_ij (numpy
//          emulation and GPU test) -- the reference's own np.linalg.norm (OpenBLAS sdot, fp32 accumulation) is
//          1e-6 .. 8e-5 on the same data.  Half the MFMAs and two thirds of the operand bytes of bf16x3.  The scale is
//          undone exactly (power of two, in fp64) when a chunk's sum enters the slab.
//   bf16x3 EXACTLY gram.hip's split mode, operation for operation (three truncated bf16 planes, six terms in the same
//          order per 16-column step, 256-column MFMA chains, fp32 level-1 sums, fp64 slab per chunk): the Gram is bitwise
//          the fused kernel's; tests/test_gpu_scale.py::test_plane_gram_is_bitwise_the_fused_gram holds it to that.
//
// The super-chunk SC (a multiple of the 8192-column chunk) is sized by a memory budget; per super-chunk one split launch
// and one tile launch, stream-ordered.  Inside a launch the schedule is gram.hip's chunked one: chunks of 8192 columns,
// all chunks of a tile add in chunk order into the tile's fp64 slabs (ticket per tile), XCD-partitioned tile list in
// super-block order, rounds that start together.
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace byz {
namespace {

constexpr int kWgRows = 256;            // rows of the workgroup tile (A side)
constexpr int kWgCols = 128;            // columns of the workgroup tile (B side) = one slab edge
constexpr int kSlab = 128;              // slab edge of gram.hip (TM)
constexpr int kThreads = 512;
constexpr int kFragBytes = 1024;        // 64 lanes x 16 bytes
constexpr int kRowBlocks = (kWgRows + kWgCols) / 32;  // 8 A + 4 B
constexpr int kChunkCols = 8192;        // must match gram.hip's chunk (the fp32 level-1 sums never spill inside one)
constexpr int kChunkSteps = kChunkCols / 16;
constexpr int kFlushSteps = 16;         // 256-column MFMA chains (the 16-bit MFMAs accumulate with truncation)
constexpr int kStatusLostTicket = 1;
constexpr int kDefaultSpan = 1;        // chunks a workgroup of the deferred tile kernel walks (BYZ_GRAM_KSPAN; round 6 A/B)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// gram.hip's split8, verbatim in effect: x = h + m + l, the three 8-bit fields of the 24-bit significand (truncation)
__device__ __forceinline__ void split_bf16x3(const f32x4& lo4, const f32x4& hi4, u32x4& hp, u32x4& mp, u32x4& lp) {
    uint32_t xb[8], r1b[8], r2b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? lo4[e] : hi4[e - 4];
        xb[e] = __float_as_uint(x);
        const float r1 = x - __uint_as_float(xb[e] & 0xffff0000u);
        r1b[e] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(r1b[e] & 0xffff0000u);
        r2b[e] = __float_as_uint(r2);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hp[e] = __builtin_amdgcn_perm(xb[2 * e + 1], xb[2 * e], 0x07060302u);
        mp[e] = __builtin_amdgcn_perm(r1b[2 * e + 1], r1b[2 * e], 0x07060302u);
        lp[e] = __builtin_amdgcn_perm(r2b[2 * e + 1], r2b[2 * e], 0x07060302u);
    }
}

// x * scale = h + m + r, h and m fp16 (round to nearest), |r| <= 2^-23 |x * scale| (x - h is exact in fp32)
__device__ __forceinline__ void split_f16x2(const f32x4& lo4, const f32x4& hi4, float scale, u32x4& hp, u32x4& mp) {
    f16x8 h, m;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = (e < 4 ? lo4[e] : hi4[e - 4]) * scale;
        h[e] = static_cast<_Float16>(x);
        m[e] = static_cast<_Float16>(x - static_cast<float>(h[e]));
    }
    hp = __builtin_bit_cast(u32x4, h);
    mp = __builtin_bit_cast(u32x4, m);
}

__device__ __forceinline__ f32x4 load4_tail(const float* __restrict__ src, int64_t k, int64_t n_cols) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (k + 3 < n_cols) {
        v = *reinterpret_cast<const f32x4*>(src + k);
    } else {
        if (k + 0 < n_cols) v.x = src[k + 0];
        if (k + 1 < n_cols) v.y = src[k + 1];
        if (k + 2 < n_cols) v.z = src[k + 2];
    }
    return v;
}

// bf16x3: one workgroup = a 32-row block x 128 columns (8 steps).  Reads are 128-byte row segments, the transposition
// into fragment order goes through LDS, writes are whole 1 KiB fragment images.
constexpr int kSplitCols = 128;
__global__ __launch_bounds__(256) void plane_split_bf16_kernel(const float* __restrict__ G, int64_t n_rows, int64_t n_cols,
                                                               int64_t ld, const int32_t* __restrict__ row_index,
                                                               int64_t k0, int64_t n_steps, u32x4* __restrict__ planes) {
    __shared__ __attribute__((aligned(16))) float tile[32][kSplitCols + 4];
    const int tid = threadIdx.x;
    const int64_t rb = blockIdx.y;
    const int64_t step0 = static_cast<int64_t>(blockIdx.x) * (kSplitCols / 16);
    {
        const int r = tid >> 3, c = tid & 7;
        int64_t row = rb * 32 + r;
        if (row > n_rows - 1) row = n_rows - 1;     // rows past the matrix: clamped, they land in Gram entries nobody reads
        if (row_index != nullptr) row = row_index[row];
        const float* src = G + row * ld;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int col = 4 * c + 32 * p;
            *reinterpret_cast<f32x4*>(&tile[r][col]) = load4_tail(src, k0 + step0 * 16 + col, n_cols);
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = 2 * wave + i;
        const int64_t step = step0 + sl;
        if (step >= n_steps) continue;
        const int row = lane & 31, kk = sl * 16 + 8 * (lane >> 5);
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(&tile[row][kk]);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(&tile[row][kk + 4]);
        u32x4 h, m, l;
        split_bf16x3(lo4, hi4, h, m, l);
        u32x4* out = planes + ((rb * n_steps + step) * 3) * 64 + lane;
        out[0] = h;
        out[64] = m;
        out[128] = l;
    }
}

// f16x2: one workgroup = a 32-row block x one 8192-column chunk: first the rows' largest magnitudes over the chunk (the
// 1 MiB block then sits in L2), then the split pass over it.  unscale[chunk][row] = 2^-shift, the factor that undoes the
// row's scale (a power of two: exact).
__global__ __launch_bounds__(256) void plane_split_f16_kernel(const float* __restrict__ G, int64_t n_rows, int64_t n_cols,
                                                              int64_t ld, const int32_t* __restrict__ row_index, int64_t k0,
                                                              int64_t n_steps, u32x4* __restrict__ planes,
                                                              double* __restrict__ unscale, int64_t rows_pad,
                                                              const int32_t* __restrict__ redo) {
    __shared__ __attribute__((aligned(16))) float tile[32][kSplitCols + 4];
    __shared__ float row_scale[32];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // redo != nullptr: the (chunk, row block) pairs the streaming kernel listed (redo[0] of them), dealt to the workgroups of
    // the launch in a grid-stride loop: however many were listed, they are all redone and the host never has to read the
    // count back (the read-back serialised host and device once per super-chunk at N = 10,000)
    const int listed = redo != nullptr ? redo[0] : 1;
    for (int item = redo != nullptr ? static_cast<int>(blockIdx.x) : 0; item < listed; item += static_cast<int>(gridDim.x)) {
    const int64_t rb = redo != nullptr ? redo[2 + 2 * item] : blockIdx.y;
    const int64_t chunk = redo != nullptr ? redo[1 + 2 * item] : blockIdx.x;
    const int r = tid >> 3, c = tid & 7;
    int64_t row = rb * 32 + r;
    if (row > n_rows - 1) row = n_rows - 1;
    if (row_index != nullptr) row = row_index[row];
    const float* src = G + row * ld;
    const int64_t kbeg = k0 + chunk * kChunkCols;
    {
        float mx = 0.0f;
        bool bad = false;
#pragma unroll 8
        for (int i = 0; i < kChunkCols / 32; ++i) {
            const f32x4 v = load4_tail(src, kbeg + 4 * c + 32 * i, n_cols);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __builtin_fabsf(v[e]);
                bad = bad || !(a <= 3.0e38f);          // inf or NaN: no scaling, the poison propagates as it is
                mx = __builtin_fmaxf(mx, a);
            }
        }
#pragma unroll
        for (int msk = 1; msk < 8; msk <<= 1) {
            mx = __builtin_fmaxf(mx, __shfl_xor(mx, msk, 64));
            bad = bad || (__shfl_xor(bad ? 1 : 0, msk, 64) != 0);
        }
        int shift = 0;
        if (!bad && mx > 0.0f) {
            shift = 14 - (static_cast<int>((__float_as_uint(mx) >> 23) & 0xffu) - 127);   // mx 2^shift in [2^14, 2^15)
            if (mx < 1.17549435e-38f) shift = 126;   // subnormal magnitudes: as far up as a float scale goes
            if (shift > 126) shift = 126;
            if (shift < -126) shift = -126;
        }
        if (c == 0) {
            row_scale[r] = __uint_as_float(static_cast<uint32_t>(shift + 127) << 23);
            unscale[chunk * rows_pad + rb * 32 + r] = __longlong_as_double(static_cast<long long>(1023 - shift) << 52);
        }
    }
    __syncthreads();
    const int64_t step_base = chunk * kChunkSteps;
    for (int sub = 0; sub < kChunkCols / kSplitCols; ++sub) {
        const int64_t step0 = step_base + sub * (kSplitCols / 16);
        if (step0 >= n_steps) break;   // uniform
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int col = 4 * c + 32 * p;
            *reinterpret_cast<f32x4*>(&tile[r][col]) = load4_tail(src, k0 + step0 * 16 + col, n_cols);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sl = 2 * wave + i;
            const int64_t step = step0 + sl;
            if (step >= n_steps) continue;
            const int frow = lane & 31, kk = sl * 16 + 8 * (lane >> 5);
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(&tile[frow][kk]);
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(&tile[frow][kk + 4]);
            u32x4 h, m;
            split_f16x2(lo4, hi4, row_scale[frow], h, m);
            u32x4* out = planes + ((rb * n_steps + step) * 2) * 64 + lane;
            out[0] = h;
            out[64] = m;
        }
        __syncthreads();
    }
    if (redo == nullptr) break;
    __syncthreads();
    }
}

// f16x2 in ONE pass over G (round 3).  The two-pass kernel above reads every element twice -- once for the row's largest
// magnitude over the chunk, once to split -- and the second read does not come out of L2 (the resident workgroups hold far
// more than 4 MiB of 1 MiB blocks): 320 GB read + 164 GB written per 160 GB of G, at copy speed.
// The scale does not have to put the chunk's largest magnitude into [2^14, 2^15): fp16 rounding is invariant under a power
// of two as long as nothing overflows and the large elements' low plane stays normal, i.e. as long as
//     2^7 <= (largest magnitude) * 2^shift < 2^16.
// So the shift comes from a SAMPLE of the row's chunk (16 groups of 8 columns, every 512 columns: 512 of its 32,768 bytes)
// placed at [2^9, 2^10) -- six binades of head room above, two below -- the chunk is then split as it streams by, and the
// true largest magnitude, which falls out of the same pass, says whether the assumption held.  Where it did not (an outlier
// 64 times the sampled maximum, a chunk whose sampled columns are all zero, inf / NaN) the (row block, chunk) goes on a
// list and the two-pass kernel redoes exactly those.  For every other block the planes are what the exact scale would give,
// scaled by a power of two that `unscale` undoes -- bit for bit for every element whose two planes stay NORMAL fp16 numbers
// under both scales; an element below ~2^-17 of the row's largest magnitude lands on a different subnormal grid (its planes
// differ by ~2^-32 of that magnitude: harmless numerically, but the Gram under the sampled scale is NOT bitwise the two-pass
// Gram, and twin rows in row blocks of which only one was redone do not give bitwise equal Gram entries -- which is why
// rows proven identical get their zero distance from the proof, gram.hip distance_kernel, not from cancellation).
constexpr int kSampleGroups = 16;
__global__ __launch_bounds__(256) void plane_split_f16_stream_kernel(const float* __restrict__ G, int64_t n_rows, int64_t n_cols,
                                                                     int64_t ld, const int32_t* __restrict__ row_index, int64_t k0,
                                                                     int64_t n_steps, u32x4* __restrict__ planes,
                                                                     double* __restrict__ unscale, int64_t rows_pad,
                                                                     int32_t* __restrict__ redo) {
    __shared__ __attribute__((aligned(16))) float tile[2][32][kSplitCols + 4];
    __shared__ float row_scale[32];
    __shared__ int block_bad;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t rb = blockIdx.y;
    const int64_t chunk = blockIdx.x;
    const int r = tid >> 3, c = tid & 7;
    int64_t row = rb * 32 + r;
    if (row > n_rows - 1) row = n_rows - 1;
    if (row_index != nullptr) row = row_index[row];
    const float* src = G + row * ld;
    const int64_t kbeg = k0 + chunk * kChunkCols;
    if (tid == 0) block_bad = 0;
    // ---- the sample: thread c of the row's eight takes groups 2 c and 2 c + 1 (eight columns each)
    int shift = 0;
    {
        float mx = 0.0f;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int64_t k = kbeg + (2 * c + g) * (kChunkCols / kSampleGroups);
            const f32x4 a = load4_tail(src, k, n_cols), b = load4_tail(src, k + 4, n_cols);
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(a[e]), __builtin_fabsf(b[e])));
        }
#pragma unroll
        for (int msk = 1; msk < 8; msk <<= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, msk, 64));
        // (NaN never wins an fmax and inf gives an out-of-range exponent: both are caught by the streaming pass)
        if (mx >= 1.17549435e-38f && mx <= 3.0e38f) {
            shift = 9 - (static_cast<int>((__float_as_uint(mx) >> 23) & 0xffu) - 127);   // mx 2^shift in [2^9, 2^10)
            if (shift > 126) shift = 126;
            if (shift < -126) shift = -126;
        }
        if (c == 0) row_scale[r] = __uint_as_float(static_cast<uint32_t>(shift + 127) << 23);
    }
    // ---- the stream: sub-block s + 1 is in flight while sub-block s is transposed through LDS and split
    const int64_t step_base = chunk * kChunkSteps;
    const float scale = __uint_as_float(static_cast<uint32_t>(shift + 127) << 23);
    float true_max = 0.0f;
    bool bad = false;
    f32x4 cur[4], nxt[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) cur[p] = load4_tail(src, k0 + step_base * 16 + 4 * c + 32 * p, n_cols);
    __syncthreads();   // row_scale
    for (int sub = 0; sub < kChunkCols / kSplitCols; ++sub) {
        const int64_t step0 = step_base + sub * (kSplitCols / 16);
        if (step0 >= n_steps) break;   // uniform
        const bool more = sub + 1 < kChunkCols / kSplitCols && step0 + kSplitCols / 16 < n_steps;
        if (more) {
#pragma unroll
            for (int p = 0; p < 4; ++p) nxt[p] = load4_tail(src, k0 + (step0 + kSplitCols / 16) * 16 + 4 * c + 32 * p, n_cols);
        }
        float (*t)[kSplitCols + 4] = tile[sub & 1];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(&t[r][4 * c + 32 * p]) = cur[p];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __builtin_fabsf(cur[p][e]);
                bad = bad || !(a <= 3.0e38f);
                true_max = __builtin_fmaxf(true_max, a);
            }
        }
        __syncthreads();   // one barrier per sub-block: the other tile is free again once everybody has passed this one
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int sl = 2 * wave + i;
            const int64_t step = step0 + sl;
            if (step >= n_steps) continue;
            const int frow = lane & 31, kk = sl * 16 + 8 * (lane >> 5);
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(&t[frow][kk]);
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(&t[frow][kk + 4]);
            u32x4 h, m;
            split_f16x2(lo4, hi4, row_scale[frow], h, m);
            u32x4* out = planes + ((rb * n_steps + step) * 2) * 64 + lane;
            out[0] = h;
            out[64] = m;
        }
        if (more) {
#pragma unroll
            for (int p = 0; p < 4; ++p) cur[p] = nxt[p];
        }
    }
    // ---- did the sampled scale hold?  2^7 <= max 2^shift < 2^16 (an all-zero chunk holds with any scale)
#pragma unroll
    for (int msk = 1; msk < 8; msk <<= 1) {
        true_max = __builtin_fmaxf(true_max, __shfl_xor(true_max, msk, 64));
        bad = bad || (__shfl_xor(bad ? 1 : 0, msk, 64) != 0);
    }
    const float scaled = true_max * scale;
    const bool holds = !bad && (true_max == 0.0f || (scaled >= 128.0f && scaled < 65536.0f));
    if (c == 0) {
        unscale[chunk * rows_pad + rb * 32 + r] = __longlong_as_double(static_cast<long long>(1023 - shift) << 52);
        if (!holds) block_bad = 1;
    }
    __syncthreads();
    if (tid == 0 && block_bad != 0) {
        const int at = atomicAdd(redo, 1);
        redo[1 + 2 * at] = static_cast<int32_t>(chunk);
        redo[2 + 2 * at] = static_cast<int32_t>(rb);
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {   // n is wave-uniform; only the first and last stages of a chunk
    switch (n) {
#define BYZ_W(k) case k: wait_vmcnt<k>(); break;
        BYZ_W(0) BYZ_W(3) BYZ_W(4) BYZ_W(5) BYZ_W(6) BYZ_W(8) BYZ_W(9) BYZ_W(10) BYZ_W(12) BYZ_W(15) BYZ_W(16) BYZ_W(18) BYZ_W(20) BYZ_W(24)
#undef BYZ_W
        default: wait_vmcnt<0>(); break;
    }
}

template <int PLANES, int MB>
struct Frags;
template <int MB>
struct Frags<3, MB> {
    typedef bf16x8 frag_t;
    bf16x8 a[3][MB], b[3][2];   // [plane h, m, l][block]
};
template <int MB>
struct Frags<2, MB> {
    typedef f16x8 frag_t;
    f16x8 a[2][MB], b[2][2];    // [plane h, m][block]
};

// PLANES = 3: bf16x3 (gram.hip's arithmetic), 2: f16x2.  NBUF stages of LDS; the DMA runs NBUF stages ahead.
// DBG (timing experiments only, wrong results; scripts/gram_ab.py): bit 0 no DMA after the first NBUF stages,
// bit 1 no MFMA, bit 2 no workgroup barrier in the steady loop, bit 3 no LDS reads after stage 0, bit 4 no slab update,
// bit 5 every workgroup's DMA reads tile (0, 0) (the L2 -> LDS rate without misses).
// MB = 32-row blocks of a wave's sub-tile along the A side: 2 -> eight waves of 64 x 64, two per SIMD.  (MB = 4, four waves of
// 128 x 64, was built in round 4, is bitwise equal and lost by 8.6 %: EXPERIMENTS.md G4; only MB = 2 is instantiated.)
// DEFER (round 5, f16x2): a chunk does not read-modify-write its tile's fp64 slabs (512 KB through the fabric per workgroup
// and chunk, behind a ticket, with the matrix pipe idle: 8 % of the kernel, EXPERIMENTS.md G3) but WRITES its level-1 sums as
// they are -- fp32, 128 KB, no read, no ticket -- into `chunk_sums`; chunk_reduce_kernel adds them into the slabs afterwards,
// in chunk order, in the same fp64 operations: the Gram is bitwise the same.
template <int PLANES, int NBUF, int DBG, int MB = 2, bool DEFER = false>
__global__ __launch_bounds__(64 * 16 / MB, 1) void gram_planes_kernel(const u32x4* __restrict__ planes, int64_t n_steps,
                                                                  const double* __restrict__ unscale, int64_t rows_pad,
                                                                  double* __restrict__ partial, int n_tiles,
                                                                  const int2* __restrict__ tile_order, int n_chunks,
                                                                  int* __restrict__ tickets, int round_size, int t128,
                                                                  int slab_live0, int n_blocks32,
                                                                  int32_t* __restrict__ device_status,
                                                                  float* __restrict__ chunk_sums,
                                                                  float* __restrict__ ragged_sums, int kspan, int pin, int map_round) {
    (void)pin;   // (gram_planes16_kernel's knob: one launch signature for both kernels)
    constexpr int kRbBytes = PLANES * kFragBytes;           // one 32-row block, one stage: [plane][1 KiB]
    constexpr int kStage = kRowBlocks * kRbBytes;           // 36,864 (bf16x3) / 24,576 (f16x2)
    constexpr int NW = 16 / MB;                             // waves of the workgroup
    constexpr int kPieces = kRowBlocks * PLANES;            // 1 KiB pieces per stage
    constexpr int kPerWave = (kPieces + NW - 1) / NW;       // DMA instructions per wave and stage (the last may be idle)
    constexpr int kFullWaves = kPieces % NW == 0 ? NW : kPieces % NW;   // waves that issue all kPerWave
    constexpr int kLead = NBUF;
    typedef Frags<PLANES, MB> frags_t;
    typedef typename frags_t::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [NBUF][12 row blocks][planes][1 KiB]

    // workgroup -> (tile, chunk): XCD x owns a contiguous share of the tile list and works through it chunk by chunk
    const int xcd = blockIdx.x & 7;
    const int seq = blockIdx.x >> 3;
    const int base = n_tiles >> 3, rem = n_tiles & 7;
    const int mine = base + (xcd < rem ? 1 : 0);
    const int first = xcd * base + (xcd < rem ? xcd : rem);
    // (DEFER, round 6) a workgroup owns a tile over a SPAN of `kspan` consecutive chunks: the planes of a row block are
    // contiguous along K, so the DMA ring runs on across the chunk boundaries -- one launch, one ramp, one drain per span -- and
    // at every boundary the chunk's level-1 sums leave for `chunk_sums` exactly as a one-chunk workgroup writes them: the Gram
    // is bitwise the same for every kspan.  `chunk` below is the FIRST chunk of the span.
    const int n_spans = DEFER ? (n_chunks + kspan - 1) / kspan : n_chunks;
    const int span = mine > 0 ? seq / mine : n_spans;
    const int chunk = DEFER ? span * kspan : span;
    int* done = tickets + n_tiles + xcd;
    {
        // rounds: a workgroup starts only when every workgroup of the earlier rounds of its XCD has finished, so that the
        // members of a round walk K in step and hit each other's operand lines in the XCD's L2 (gram.hip)
        const int round = round_size > 0 ? seq / round_size : 0;
        if (round > 0 && threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * round_size) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 20)) break;   // only speed depends on the gate
            }
        }
        __syncthreads();
        if (span >= n_spans) {
            if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    const int t_list = first + (seq - span * mine);
    const int2 tt = tile_order[t_list];
    const int bi = __builtin_amdgcn_readfirstlane(tt.x);   // 256-row block of the A side
    const int tj = __builtin_amdgcn_readfirstlane(tt.y);   // 128-row block of the B side

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ti = 2 * bi + (wr * MB) / 4;             // this wave's slab row
    const int row_in_slab = ((wr * MB) % 4) * 32;      // and where its sub-tile starts inside it
    // Which of the wave's MB x 2 output blocks (32 x 32) anybody reads: block (m, n) covers 32-row block `rblk` of the A side and
    // `cblk` of the B side.  Not needed: blocks of rows past the matrix (N = 4000 pads to 4096: three quarters of the last slab
    // row; N = 10,000 to 10,112), and blocks strictly above the diagonal (gram_reduce_kernel reads a diagonal slab's lower
    // triangle only).  Round 4 skipped at slab granularity only and issued 8 % more MFMAs than 3 N^2 D / 32768 at N = 4000
    // (VERDICT r4, weak 5).  A skipped block is never multiplied, never flushed and never touches its slab entries; the blocks
    // that are computed go through the same MFMA chain in the same order as before: the Gram is bitwise the same.
    unsigned live_blocks = 0;
    if (tj <= ti && ti < t128 && n_blocks32 < 0) {
        live_blocks = (1u << (MB * 2)) - 1u;     // BYZ_GRAM_BLOCK_SKIP=0: round 4's slab-granular rule, for the same-box A/B
    } else if (tj <= ti && ti < t128) {     // the upper half of a tile that straddles the diagonal is not needed
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int rblk = ti * 4 + row_in_slab / 32 + m, cblk = tj * 4 + wc * 2 + n;
                if (rblk < n_blocks32 && cblk <= rblk) live_blocks |= 1u << (m * 2 + n);
            }
    }
    // The K loop is instantiated once per mask that can occur (MB = 2, bit m * 2 + n): all four blocks; the row block m = 0
    // only (the last valid row block of the matrix is the first of its wave); a diagonal 64 x 64 sub-tile without its upper
    // block; both; nothing (a dead wave still takes part in the DMA and the barriers).  A run-time test per MFMA instead cost the
    // compiler its schedule (a wait for ALL pending LDS reads in front of every MFMA).
    static_assert(MB == 2, "the mask set below is the one of 64 x 64 wave tiles");
    const bool live_wave = live_blocks != 0;

    const int step0 = chunk * kChunkSteps;
    int n_stages = static_cast<int>(n_steps) - step0;
    {
        const int span_steps = DEFER ? kspan * kChunkSteps : kChunkSteps;
        if (n_stages > span_steps) n_stages = span_steps;
    }

    // LDS-DMA: piece q = wave + 8 i of the stage (row block q / PLANES, plane q % PLANES): a wave-uniform byte offset from
    // `planes` plus 16 bytes per lane; the LDS image of a stage is piece-linear, and so is a row block's stage in HBM
    int64_t piece_off[kPerWave];
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        int q = wave + NW * i;
        if (q >= kPieces) q = kPieces - 1;
        const int rbl = q / PLANES, piece = q % PLANES;
        int64_t rb = rbl < 8 ? static_cast<int64_t>(bi) * 8 + rbl : static_cast<int64_t>(tj) * 4 + (rbl - 8);
        if (DBG & 32) rb = rbl < 8 ? rbl : rbl - 8;
        piece_off[i] = (((rb * n_steps + step0) * PLANES + piece) * 64) * 16;
    }
    const unsigned char* lane_base = reinterpret_cast<const unsigned char*>(planes) + lane * 16;
    const int my_dmas = wave < kFullWaves ? kPerWave : kPerWave - 1;
    auto dma = [&](int s) __attribute__((always_inline)) {
        if ((DBG & 1) && s >= NBUF) return;
        unsigned char* dst = lds + (s % NBUF) * kStage + wave * kFragBytes;
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            if (kFullWaves != NW && i == kPerWave - 1 && wave >= kFullWaves) break;   // wave-uniform
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(lane_base + piece_off[i] + static_cast<int64_t>(s) * (PLANES * kFragBytes)),
                (__attribute__((address_space(3))) void*)(dst + i * NW * kFragBytes), 16, 0, 0);
        }
    };

    f32x16 acc[MB][2], acc2[MB][2];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[m][n][e] = 0.0f;
                acc2[m][n][e] = 0.0f;
            }

    auto read_frags = [&](int s, frags_t& f, auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;     // only the fragments a live block multiplies
        if ((DBG & 8) && s > 0) return;
        const unsigned char* A = lds + (s % NBUF) * kStage + (MB * wr) * kRbBytes + lane * 16;
        const unsigned char* B = lds + (s % NBUF) * kStage + (8 + 2 * wc) * kRbBytes + lane * 16;
#pragma unroll
        for (int p = 0; p < PLANES; ++p) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
                if (((MASK >> (m * 2)) & 3u) != 0) f.a[p][m] = *reinterpret_cast<const frag_t*>(A + m * kRbBytes + p * kFragBytes);
#pragma unroll
            for (int n = 0; n < 2; ++n)
                if (((MASK >> n) & 5u) != 0) f.b[p][n] = *reinterpret_cast<const frag_t*>(B + n * kRbBytes + p * kFragBytes);
        }
    };
    // TERMS of the split product, smallest first, term-major over the accumulators (bf16x3: the order of gram.hip's split mode)
    auto mfma_term = [&](const frags_t& f, int t, int m, int n) __attribute__((always_inline)) {
        if constexpr (PLANES == 3) {
            // h h' + h m' + m h' + m m' + h l' + l h'
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0};
            constexpr int pb[6] = {0, 2, 1, 0, 1, 0};
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[pa[t]][m], f.b[pb[t]][n], acc[m][n], 0, 0, 0);
        } else {
            constexpr int pa[3] = {1, 0, 0};   // m h' + h m' + h h'
            constexpr int pb[3] = {0, 1, 0};
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[pa[t]][m], f.b[pb[t]][n], acc[m][n], 0, 0, 0);
        }
    };
    constexpr int kTerms = PLANES == 3 ? 6 : 3;
    auto multiply = [&](const frags_t& f, auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;
#pragma unroll
        for (int t = 0; t < kTerms; ++t)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if (((MASK >> (m * 2 + n)) & 1u) != 0) mfma_term(f, t, m, n);   // folds: the loops are unrolled
    };
    // this wave's level-1 sums of chunk `c` -> chunk_sums (slab `sel` of workgroup tile t_list; element (i, j) where the fp64
    // slab has it); only the blocks somebody reads
    auto store_chunk = [&](int c, auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;
        const int sel = (wr * MB) / 4;
        // The addresses are formed HERE, from values the compiler cannot see through: hoisted out of the K loop they cost 52
        // registers and 248 bytes of scratch per lane in the hot loop (256 VGPRs against 204).
        int lane_o = lane, c_o = c;
        asm volatile("" : "+v"(lane_o));
        asm volatile("" : "+s"(c_o));
        float* out = chunk_sums + ((static_cast<int64_t>(c_o) * n_tiles + t_list) * 2 + sel) * (kSlab * kSlab) +
                     (row_in_slab + 4 * (lane_o >> 5)) * kSlab + wc * 64 + (lane_o & 31);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (((MASK >> (m * 2 + n)) & 1u) == 0) continue;
#pragma unroll
                for (int e = 0; e < 16; ++e)      // element (i, j): i = row_in_slab + 32 m + (e & 3) + 8 (e >> 2) + 4 (lane >> 5), j = 64 wc + 32 n + (lane & 31)
                    out[(m * 32 + (e & 3) + 8 * (e >> 2)) * kSlab + n * 32] = acc2[m][n][e];
            }
    };
    auto flush = [&](int s, auto mask_c) __attribute__((always_inline)) {
        if ((s + 1) % kFlushSteps == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc2[m][n][e] += acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
            if constexpr (DEFER) {
                // a chunk of the span ends here and another follows (the span's last chunk leaves in the epilogue)
                if ((s + 1) % kChunkSteps == 0 && s + 1 < n_stages) {
                    store_chunk(chunk + (s + 1) / kChunkSteps - 1, mask_c);
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc2[m][n][e] = 0.0f;
                    // stores count in vmcnt like the DMA's loads, and the counted waits of the ring assume that whatever is
                    // outstanding returns in issue order: nothing of either kind stays in flight across the boundary
                    wait_vmcnt<0>();
                }
            }
        }
    };
    // one stage: multiply `cur` (stage s, in registers), read stage s + 1 into `next`, keep the DMA kLead stages ahead
    auto steady = [&](int s, const frags_t& cur, frags_t& next, auto mask_c) __attribute__((always_inline)) {
        constexpr bool kLive = decltype(mask_c)::value != 0;
        dma(s + kLead);                       // into the buffer of stage s, whose last readers passed the previous barrier
        if (kLive && !(DBG & 2)) {
            read_frags(s + 1, next, mask_c);
            multiply(cur, mask_c);
        }
        // before anyone reads stage s + 2 it must have landed: of my requests only stages s + 3 .. s + kLead may be
        // outstanding (memory reads return in order, so that is a vmcnt bound)
        if (kFullWaves == NW || wave < kFullWaves) wait_vmcnt<(kLead - 2) * kPerWave>();
        else wait_vmcnt<(kLead - 2) * (kPerWave - 1)>();
        if (DBG & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kLive) flush(s, mask_c);
    };
    auto drain = [&](int s, const frags_t& cur, frags_t& next, auto mask_c) __attribute__((always_inline)) {
        constexpr bool kLive = decltype(mask_c)::value != 0;
        if (s + kLead < n_stages) dma(s + kLead);
        if (kLive && !(DBG & 2)) {
            if (s + 1 < n_stages) read_frags(s + 1, next, mask_c);
            multiply(cur, mask_c);
        }
        const int last = s + kLead < n_stages ? s + kLead : n_stages - 1;
        const int ahead = last - (s + 2);
        wait_vmcnt_dyn((ahead > 0 ? ahead : 0) * my_dmas);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kLive) flush(s, mask_c);
    };

    for (int a = 0; a < kLead && a < n_stages; ++a) dma(a);
    {
        const int issued = n_stages < kLead ? n_stages : kLead;
        wait_vmcnt_dyn((issued > 2 ? issued - 2 : 0) * my_dmas);   // stages 0 and 1 have landed
    }
    asm volatile("s_barrier" ::: "memory");
    auto k_loop = [&](auto mask_c) __attribute__((always_inline)) {
        constexpr bool kLive = decltype(mask_c)::value != 0;
        frags_t x, y;
        if (kLive && !(DBG & 2) && n_stages > 0) read_frags(0, x, mask_c);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // everybody has read stage 0: its buffer may be refilled
        int s = 0;
        for (; s + 1 + kLead < n_stages; s += 2) {
            steady(s, x, y, mask_c);
            steady(s + 1, y, x, mask_c);
        }
        for (; s < n_stages; s += 2) {
            drain(s, x, y, mask_c);
            if (s + 1 < n_stages) drain(s + 1, y, x, mask_c);
        }
    };
    // wave-uniform; every instantiation passes the same barriers.  A mask outside the set (there is none) computes all four.
    switch (live_blocks) {
        case 0u: k_loop(std::integral_constant<unsigned, 0u>{}); break;
        case 1u: k_loop(std::integral_constant<unsigned, 1u>{}); break;
        case 3u: k_loop(std::integral_constant<unsigned, 3u>{}); break;
        case 13u: k_loop(std::integral_constant<unsigned, 13u>{}); break;
        default: live_blocks = 15u; k_loop(std::integral_constant<unsigned, 15u>{}); break;
    }
    auto block_live = [&](int m, int n) __attribute__((always_inline)) { return ((live_blocks >> (m * 2 + n)) & 1u) != 0; };

    if constexpr (DEFER) {
        // this chunk's level-1 sums, as they are: slab `sel` of workgroup tile t_list, chunk `chunk`; element (i, j) where the
        // fp64 slab has it.  A chunk whose last MFMA chain has not been flushed (n_stages not a multiple of kFlushSteps: only
        // the ragged last chunk of the matrix) leaves that chain's sums next to them.
        if (live_wave) {
            const int sel = (wr * MB) / 4;
            const bool unflushed = (n_stages % kFlushSteps) != 0;
            const int last_chunk = chunk + (n_stages - 1) / kChunkSteps;
            float* out = chunk_sums + ((static_cast<int64_t>(last_chunk) * n_tiles + t_list) * 2 + sel) * (kSlab * kSlab);
            float* rag = ragged_sums + (static_cast<int64_t>(t_list) * 2 + sel) * (kSlab * kSlab);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    if (!block_live(m, n)) continue;
                    const int j = wc * 64 + n * 32 + (lane & 31);
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = row_in_slab + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        out[i * kSlab + j] = acc2[m][n][e];
                        if (unflushed) rag[i * kSlab + j] = acc[m][n][e];
                    }
                }
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // epilogue: chunks of a tile add into its slabs in chunk order (gram.hip's ticket protocol)
    bool lost = false;
    if (chunk > 0) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(tickets + t_list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chunk) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 26)) {   // bounded: a lost ticket must not hang the device ...
                    lost = true;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        lost = __syncthreads_or(lost ? 1 : 0) != 0;
    }
    if (lost) {
        // ... and must not be papered over: without the ticket the slab is not ours to update (the host turns the
        // status word into BYZ_E_HIP)
        if (tid == 0) atomicOr(device_status, kStatusLostTicket);
    } else if (live_wave && (DBG & 16)) {
        float all = 0.0f;   // keeps the accumulators alive without the slab traffic
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) all += acc[m][n][e] + acc2[m][n][e];
        if (all == 1.2345e38f) partial[tid] = all;
    } else if (live_wave) {
        const bool slab_live = chunk > 0 || slab_live0 != 0;
        double* out = partial + (static_cast<int64_t>(ti) * (ti + 1) / 2 + tj) * (kSlab * kSlab);
        const double* un = PLANES == 2 ? unscale + static_cast<int64_t>(chunk) * rows_pad : nullptr;
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (!block_live(m, n)) continue;     // nobody reads this block of the slab
                const int j = wc * 64 + n * 32 + (lane & 31);
                double uj = 1.0;
                if constexpr (PLANES == 2) uj = un[static_cast<int64_t>(tj) * kSlab + j];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = row_in_slab + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    double v = static_cast<double>(acc2[m][n][e]);
                    v += static_cast<double>(acc[m][n][e]);
                    if constexpr (PLANES == 2) v *= un[static_cast<int64_t>(ti) * kSlab + i] * uj;   // powers of two: exact
                    if (slab_live) v += out[i * kSlab + j];
                    out[i * kSlab + j] = v;
                }
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(tickets + t_list, chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- the f16x2 tile kernel on v_mfma_f32_16x16x32_f16 (round 6) ------------------------------------------------------------
// Same workgroup tile (256 x 128), same wave tile (64 x 64), same planes, same LDS-DMA ring, same schedule of workgroups and
// the same fp32 level-1 / fp64 slab hierarchy as gram_planes_kernel<2, ...> above -- but the matrix instruction is the
// 16 x 16 x 32 one: per MAC it reads and writes HALF the accumulator registers and twice the operand registers of the
// 32 x 32 x 16 form, and at the socket's power cap (which is what bounds this kernel: EXPERIMENTS.md G6) that is worth 12-14 % of
// MFMA throughput on random fp16 data (scripts/ubench/mfma_shapes.hip: 1430 -> 1610 TF with the operands re-read from LDS every
// step, 1580 -> 1800 TF with them in registers; on all-zero data, where nothing is power-bound, the two shapes tie).
//
// A K = 32 step is a SUPER-STAGE: two consecutive 16-column stages of the ring.  The planes are stored as 1 KiB images of
// (32 rows x 16 columns) per plane: lane l' of an image holds row l' & 31, columns 8 (l' >> 5) .. + 7.  The A / B operand of
// the 16 x 16 x 32 MFMA wants lane l: row l & 15, k = 8 (l >> 4) .. + 7 of 32.  So lane l reads 16 bytes at
//       stage (l >> 5) of the pair,  image lane (l & 15) + 16 h + 32 ((l >> 4) & 1)        (h = which half of the 32-row block)
// -- lanes 0-31 from the first stage's buffer, lanes 32-63 from the second's, one ds_read_b128 -- and the sixteen lanes of
// every service group of ds_read_b128 land on sixteen different 16-byte slots: conflict-free, no change to the split kernels.
//
// Registers: accumulators 64 + level-1 sums 64 + fragments 96 (A of this super-stage 32, A of the next 32, the two halves of B
// 16 + 16).  The B side is walked in two halves of 32 columns: the first half's 24 MFMAs cover the reads of the second half's
// B fragments, the second half's 24 MFMAs cover the reads of the NEXT super-stage's A and first-half B fragments.  One
// workgroup barrier per super-stage (half as many as the 16-column form).
template <int NBUF, bool DEFER>
__global__ __launch_bounds__(512, 1) void gram_planes16_kernel(const u32x4* __restrict__ planes, int64_t n_steps,
                                                            const double* __restrict__ unscale, int64_t rows_pad,
                                                            double* __restrict__ partial, int n_tiles,
                                                            const int2* __restrict__ tile_order, int n_chunks,
                                                            int* __restrict__ tickets, int round_size, int t128,
                                                            int slab_live0, int n_blocks32,
                                                            int32_t* __restrict__ device_status,
                                                            float* __restrict__ chunk_sums,
                                                            float* __restrict__ ragged_sums, int kspan, int pin, int map_round) {
    constexpr int PLANES = 2;
    constexpr int kRbBytes = PLANES * kFragBytes;           // one 32-row block, one stage: [plane][1 KiB]
    constexpr int kStage = kRowBlocks * kRbBytes;           // 24,576
    constexpr int NW = 8;
    constexpr int kPerWave = kRowBlocks * PLANES / NW;      // 3 DMA instructions per wave and stage
    constexpr int kSuperFlush = kFlushSteps / 2;            // super-stages per 256-column MFMA chain
    constexpr int kSuperChunk = kChunkSteps / 2;            // super-stages per 8192-column chunk
    static_assert(kRowBlocks * PLANES % NW == 0 && NBUF == 6, "ring arithmetic below");
    typedef f16x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [NBUF][12 row blocks][2 planes][1 KiB]

    // workgroup -> (tile, span of chunks): exactly gram_planes_kernel's mapping
    const int xcd = blockIdx.x & 7;
    const int seq = blockIdx.x >> 3;
    const int base = n_tiles >> 3, rem = n_tiles & 7;
    const int mine = base + (xcd < rem ? 1 : 0);
    const int first = xcd * base + (xcd < rem ? xcd : rem);
    const int n_spans = DEFER ? (n_chunks + kspan - 1) / kspan : n_chunks;
    // map_round > 0 (round 6, the default): the (span, tile) units of the launch form ONE sequence, span-major, and the XCDs take
    // its runs of `map_round` units in turn -- every round of an XCD is `map_round` CONSECUTIVE tiles of the list on (but for one
    // round in n_tiles / map_round) ONE chunk, so with the list in bands of four 256-row blocks a round touches 16 row blocks of
    // 128: the fewest 32 tiles of 256 x 128 can.  (map_round == 0: gram_planes_kernel's mapping -- a contiguous share of the tile
    // list per XCD, whose rounds straddle two chunks nearly always.)
    // (map_round < 0: rounds of -map_round units CLAIMED from one counter as the XCDs get to them instead of dealt in turn: an XCD
    // whose rounds held the cheaper diagonal tiles takes more of them, and the launch ends with every XCD busy.)
    int span = mine > 0 ? seq / mine : n_spans;
    int t_in_share = seq - span * mine;
    const int mr = map_round < 0 ? -map_round : map_round;
    int* done = tickets + n_tiles + xcd;
    __shared__ int claimed_round;
    {
        const int round = round_size > 0 ? seq / round_size : 0;
        if (round > 0 && threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * round_size) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 20)) break;   // only speed depends on the gate
            }
        }
        if (map_round < 0 && threadIdx.x == 0) {
            // the first workgroup of this XCD's round `seq / mr` to get here claims the next run of units for all of them
            int* slot = tickets + n_tiles + 9 + xcd * static_cast<int>(gridDim.x / (8u * static_cast<unsigned>(mr))) + seq / mr;
            int v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == 0) {
                int expected = 0;
                if (__hip_atomic_compare_exchange_strong(slot, &expected, -1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    v = __hip_atomic_fetch_add(tickets + n_tiles + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                    __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    v = expected;
                }
            }
            while (v <= 0) {   // (the claimer is between its exchange and its store: a fetch-add away)
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            claimed_round = v - 1;
        }
        __syncthreads();
        if (mr > 0) {
            const int round_global = map_round < 0 ? claimed_round : (seq / mr) * 8 + xcd;
            const int unit = round_global * mr + seq % mr;
            span = unit / n_tiles;
            t_in_share = unit - span * n_tiles - first;
        }
        if (span >= n_spans) {
            if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    const int chunk = DEFER ? span * kspan : span;
    const int t_list = first + t_in_share;
    const int2 tt = tile_order[t_list];
    const int bi = __builtin_amdgcn_readfirstlane(tt.x);   // 256-row block of the A side
    const int tj = __builtin_amdgcn_readfirstlane(tt.y);   // 128-row block of the B side

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ti = 2 * bi + wr / 2;                    // this wave's slab row
    const int row_in_slab = (wr % 2) * 64;             // and where its 64 x 64 sub-tile starts inside it
    // live 32 x 32 blocks of the sub-tile (bit m32 * 2 + n32): the rule of gram_planes_kernel
    unsigned live_blocks = 0;
    if (tj <= ti && ti < t128 && n_blocks32 < 0) {
        live_blocks = 15u;
    } else if (tj <= ti && ti < t128) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int rblk = ti * 4 + row_in_slab / 32 + m, cblk = tj * 4 + wc * 2 + n;
                if (rblk < n_blocks32 && cblk <= rblk) live_blocks |= 1u << (m * 2 + n);
            }
    }
    const bool live_wave = live_blocks != 0;

    const int step0 = chunk * kChunkSteps;
    int n_stages = static_cast<int>(n_steps) - step0;       // even: n_steps counts whole 32-column pairs, chunks are even
    {
        const int span_steps = DEFER ? kspan * kChunkSteps : kChunkSteps;
        if (n_stages > span_steps) n_stages = span_steps;
    }
    const int n_super = n_stages / 2;

    // LDS-DMA: piece q = wave + 8 i of a stage (row block q / 2, plane q % 2)
    int64_t piece_off[kPerWave];
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        const int q = wave + NW * i;
        const int rbl = q / PLANES, piece = q % PLANES;
        const int64_t rb = rbl < 8 ? static_cast<int64_t>(bi) * 8 + rbl : static_cast<int64_t>(tj) * 4 + (rbl - 8);
        piece_off[i] = (((rb * n_steps + step0) * PLANES + piece) * 64) * 16;
    }
    const unsigned char* lane_base = reinterpret_cast<const unsigned char*>(planes) + lane * 16;
    auto dma_piece = [&](int s, int i) __attribute__((always_inline)) {   // (i: a compile-time piece number after unrolling)
        unsigned char* dst = lds + (s % NBUF) * kStage + wave * kFragBytes;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(lane_base + piece_off[i] + static_cast<int64_t>(s) * (PLANES * kFragBytes)),
            (__attribute__((address_space(3))) void*)(dst + i * NW * kFragBytes), 16, 0, 0);
    };
    auto dma = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) dma_piece(s, i);
    };

    f32x4 acc[4][4], acc2[4][4];     // [16-row block][16-column block]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[m][n][e] = 0.0f;
                acc2[m][n][e] = 0.0f;
            }

    // this lane's byte offset inside a 32-row block's stage image pair: which stage of the pair, which image lane
    const int lane_in_image = (lane & 15) + 32 * ((lane >> 4) & 1);
    const int pair_half = lane >> 5;                         // 0: the first stage of the super-stage, 1: the second
    // fragment (row block rb32 of the stage image, half h of it, plane p) of super-stage sp
    auto frag_at = [&](int sp, int rb32, int h, int p) __attribute__((always_inline)) -> frag_t {
        // (NBUF = 6 is even: the two stages of a super-stage sit in buffers 2 (sp % 3) and 2 (sp % 3) + 1)
        const unsigned char* at = lds + (2 * (sp % (NBUF / 2)) + pair_half) * kStage + rb32 * kRbBytes + p * kFragBytes +
                                  (lane_in_image + 16 * h) * 16;
        return *reinterpret_cast<const frag_t*>(at);
    };
    struct AFrags { frag_t v[2][4]; };     // [plane h, m][16-row block of the wave's 64 rows]
    struct BFrags { frag_t v[2][2]; };     // [plane][16-column block of one 32-column half]
    auto read_a = [&](int sp, AFrags& a, auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (((MASK >> ((m / 2) * 2)) & 3u) != 0) a.v[p][m] = frag_at(sp, 2 * wr + m / 2, m % 2, p);
    };
    auto read_b = [&](int sp, int half, BFrags& b, auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;
        // (the caller passes a compile-time `half`)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int n = 0; n < 2; ++n) b.v[p][n] = frag_at(sp, 8 + 2 * wc + half, n, p);
        (void)MASK;
    };
    // the 24 MFMAs of one 32-column half: m h' + h m' + h h' per 16 x 16 block, term-major
    auto multiply_half = [&](const AFrags& a, const BFrags& b, auto half_c, auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;
        constexpr int HALF = decltype(half_c)::value;
        constexpr int pa[3] = {1, 0, 0};
        constexpr int pb[3] = {0, 1, 0};
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    if (((MASK >> ((m / 2) * 2 + HALF)) & 1u) != 0)
                        acc[m][2 * HALF + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v[pa[t]][m], b.v[pb[t]][n], acc[m][2 * HALF + n], 0, 0, 0);
    };
    // element (i, j) of the wave's sub-tile that acc[m][n][e] holds: i = 16 m + 4 (lane >> 4) + e, j = 16 n + (lane & 15)
    auto store_chunk = [&](int c, auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;
        const int sel = wr / 2;
        int lane_o = lane, c_o = c;
        asm volatile("" : "+v"(lane_o));      // (the addresses are formed here, not hoisted out of the K loop)
        asm volatile("" : "+s"(c_o));
        float* out = chunk_sums + ((static_cast<int64_t>(c_o) * n_tiles + t_list) * 2 + sel) * (kSlab * kSlab) +
                     (row_in_slab + 4 * (lane_o >> 4)) * kSlab + wc * 64 + (lane_o & 15);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (((MASK >> ((m / 2) * 2 + n / 2)) & 1u) == 0) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[(16 * m + e) * kSlab + 16 * n] = acc2[m][n][e];
            }
    };
    auto flush = [&](int sp, auto mask_c) __attribute__((always_inline)) {
        if ((sp + 1) % kSuperFlush == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc2[m][n][e] += acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
            if constexpr (DEFER) {
                if ((sp + 1) % kSuperChunk == 0 && sp + 1 < n_super) {   // a chunk of the span ends here and another follows
                    store_chunk(chunk + (sp + 1) / kSuperChunk - 1, mask_c);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc2[m][n][e] = 0.0f;
                    wait_vmcnt<0>();   // stores count in vmcnt like the ring's loads: nothing of either kind crosses the boundary
                }
            }
        }
    };

    // ---- the ring: six stages = three super-stages; the DMA runs two super-stages ahead of the one being multiplied
    for (int a = 0; a < NBUF && a < n_stages; ++a) dma(a);
    {
        const int issued = n_stages < NBUF ? n_stages : NBUF;
        wait_vmcnt_dyn((issued > 2 ? issued - 2 : 0) * kPerWave);   // stages 0 and 1 have landed
    }
    asm volatile("s_barrier" ::: "memory");
    auto k_loop = [&](auto mask_c) __attribute__((always_inline)) {
        constexpr unsigned MASK = decltype(mask_c)::value;
        constexpr bool kLive = MASK != 0;
        constexpr std::integral_constant<int, 0> h0{};
        constexpr std::integral_constant<int, 1> h1{};
        AFrags a0, a1;
        BFrags b0, b1;
        if (kLive && n_super > 0) {
            read_a(0, a0, mask_c);
            if ((MASK & 5u) != 0) read_b(0, 0, b0, mask_c);
        }
        // one super-stage: cur A in `a`, next A goes to `an`
        auto super = [&](int sp, AFrags& a, AFrags& an) __attribute__((always_inline)) {
            if (kLive) {
                if ((MASK & 10u) != 0) read_b(sp, 1, b1, mask_c);            // the second half's B: under the first half's MFMAs
                if ((MASK & 5u) != 0) multiply_half(a, b0, h0, mask_c);
            }
            // stages 2 sp + 2, 2 sp + 3 must have landed before anybody reads them; of my requests only the stages behind them
            // may still be outstanding (returns are in issue order)
            {
                const int last = 2 * sp + 5 < n_stages - 1 ? 2 * sp + 5 : n_stages - 1;   // the last stage requested so far
                const int ahead = last - (2 * sp + 3);
                if (2 * sp + 7 < n_stages) wait_vmcnt<2 * kPerWave>();                   // steady state: two stages may be in flight
                else wait_vmcnt_dyn((ahead > 0 ? ahead : 0) * kPerWave);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everybody has read super-stage sp; sp + 1 has landed
            if (2 * sp + NBUF < n_stages) dma(2 * sp + NBUF);
            if (2 * sp + NBUF + 1 < n_stages) dma(2 * sp + NBUF + 1);
            if (kLive) {
                if (sp + 1 < n_super) {
                    read_a(sp + 1, an, mask_c);                               // under the second half's MFMAs
                    if ((MASK & 5u) != 0) read_b(sp + 1, 0, b0, mask_c);
                }
                if ((MASK & 10u) != 0) multiply_half(a, b1, h1, mask_c);
                flush(sp, mask_c);
            }
        };
        // The steady state of a FULL wave (all sixteen blocks live, both stages of super-stage sp + 3 still to be requested) with its
        // instructions in a PINNED order (BYZ_GRAM_PIN=0: the compiler's): behind the barrier six groups of {one DMA piece, two
        // fragment reads of super-stage sp + 1, four MFMAs of this super-stage's second half}.  Left to itself hipcc issues the
        // six DMA pieces and the twelve reads first -- both waves of every SIMD at once, right behind the barrier, with the
        // matrix pipe idle -- and the 24 MFMAs behind them (same box: 45.6 -> 42.3 ms per launch at N = 4000, bitwise).
        // (Three other pinned orders -- the four MFMAs in front of the DMA piece, twelve groups of {two MFMAs, a DMA piece every
        // other group, one read}, the first half pinned as well -- were alternated on one box: 41.3 .. 41.8 ms, all four:
        // profiles/r06o_gram_pin_orders_ab.txt.  What pays is that the DMA pieces and the reads are spread AT ALL.)
        // The steady state runs in TRIPS OF EIGHT super-stages = one 256-column MFMA chain, every position of the chain its own
        // straight-line code:
        //   * position 0 starts every block's chain from C = 0 (an inline constant of the MFMA), so the flush at position 7
        //     does not have to zero the accumulators: 64 of its 128 vector instructions gone;
        //   * position 7 adds the first half's eight blocks to the level-1 sums INSIDE the six pinned groups behind the barrier
        //     (their last MFMAs were issued before it) and only the second half's eight behind them;
        //   * the first half is pinned too: one read of the second half's B per six MFMAs.
        auto super8 = [&](int sp, AFrags& a, AFrags& an, auto pos_c) __attribute__((always_inline)) {
            constexpr int POS = decltype(pos_c)::value;
            constexpr bool FRESH = POS == 0, FLUSHING = POS == 7;
            constexpr int pa[3] = {1, 0, 0};
            constexpr int pb[3] = {0, 1, 0};
            const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                b1.v[g / 2][g % 2] = frag_at(sp, 8 + 2 * wc + 1, g % 2, g / 2);
#pragma unroll
                for (int q = 6 * g; q < 6 * g + 6; ++q) {
                    const int t = q / 8, m = (q % 8) / 2, n = q % 2;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v[pa[t]][m], b0.v[pb[t]][n], (FRESH && t == 0) ? zero4 : acc[m][n], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wait_vmcnt<2 * kPerWave>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                dma_piece(2 * sp + NBUF + g / 3, g % 3);
#pragma unroll
                for (int r = 2 * g; r < 2 * g + 2; ++r) {
                    if (r < 8) an.v[r / 4][r % 4] = frag_at(sp + 1, 2 * wr + (r % 4) / 2, (r % 4) % 2, r / 4);
                    else b0.v[(r - 8) / 2][(r - 8) % 2] = frag_at(sp + 1, 8 + 2 * wc + 0, (r - 8) % 2, (r - 8) / 2);
                }
#pragma unroll
                for (int q = 4 * g; q < 4 * g + 4; ++q) {
                    const int t = q / 8, m = (q % 8) / 2, n = q % 2;
                    acc[m][2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v[pa[t]][m], b1.v[pb[t]][n], (FRESH && t == 0) ? zero4 : acc[m][2 + n], 0, 0, 0);
                }
                if constexpr (FLUSHING) {
                    // blocks (m, n < 2) of the first half: b = 2 m + n, eight of them over the six groups
#pragma unroll
                    for (int b = (8 * g) / 6; b < (8 * (g + 1)) / 6; ++b)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[b / 2][b % 2][e] += acc[b / 2][b % 2][e];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (FLUSHING) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 2; n < 4; ++n)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc2[m][n][e] += acc[m][n][e];
                if constexpr (DEFER) {
                    if ((sp + 1) % kSuperChunk == 0) {        // a chunk of the span ends here and another follows (steady state)
                        store_chunk(chunk + (sp + 1) / kSuperChunk - 1, mask_c);
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int n = 0; n < 4; ++n)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc2[m][n][e] = 0.0f;
                        wait_vmcnt<0>();
                    }
                }
            }
        };
        int sp = 0;
        if constexpr (MASK == 15u) {
            if (pin != 0) {
                bool any = false;
                for (; 2 * (sp + 7) + NBUF + 1 < n_stages && sp + 8 < n_super; sp += 8) {
                    super8(sp + 0, a0, a1, std::integral_constant<int, 0>{});
                    super8(sp + 1, a1, a0, std::integral_constant<int, 1>{});
                    super8(sp + 2, a0, a1, std::integral_constant<int, 2>{});
                    super8(sp + 3, a1, a0, std::integral_constant<int, 3>{});
                    super8(sp + 4, a0, a1, std::integral_constant<int, 4>{});
                    super8(sp + 5, a1, a0, std::integral_constant<int, 5>{});
                    super8(sp + 6, a0, a1, std::integral_constant<int, 6>{});
                    super8(sp + 7, a1, a0, std::integral_constant<int, 7>{});
                    any = true;
                }
                if (any) {     // the trips leave the accumulators as the last chain left them: what follows starts from zero
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[m][n][e] = 0.0f;
                }
            }
        }
        for (; sp + 1 < n_super; sp += 2) {
            super(sp, a0, a1);
            super(sp + 1, a1, a0);
        }
        if (sp < n_super) super(sp, a0, a1);
    };
    switch (live_blocks) {
        case 0u: k_loop(std::integral_constant<unsigned, 0u>{}); break;
        case 1u: k_loop(std::integral_constant<unsigned, 1u>{}); break;
        case 3u: k_loop(std::integral_constant<unsigned, 3u>{}); break;
        case 13u: k_loop(std::integral_constant<unsigned, 13u>{}); break;
        default: live_blocks = 15u; k_loop(std::integral_constant<unsigned, 15u>{}); break;
    }
    auto block_live = [&](int m, int n) __attribute__((always_inline)) { return ((live_blocks >> ((m / 2) * 2 + n / 2)) & 1u) != 0; };

    if constexpr (DEFER) {
        if (live_wave) {
            const int sel = wr / 2;
            const bool unflushed = (n_super % kSuperFlush) != 0;
            const int last_chunk = chunk + (n_stages - 1) / kChunkSteps;
            float* out = chunk_sums + ((static_cast<int64_t>(last_chunk) * n_tiles + t_list) * 2 + sel) * (kSlab * kSlab);
            float* rag = ragged_sums + (static_cast<int64_t>(t_list) * 2 + sel) * (kSlab * kSlab);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (!block_live(m, n)) continue;
                    const int j = wc * 64 + 16 * n + (lane & 15);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = row_in_slab + 16 * m + 4 * (lane >> 4) + e;
                        out[i * kSlab + j] = acc2[m][n][e];
                        if (unflushed) rag[i * kSlab + j] = acc[m][n][e];
                    }
                }
        }
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // epilogue of the in-kernel form: chunks of a tile add into its slabs in chunk order (gram.hip's ticket protocol)
    bool lost = false;
    if (chunk > 0) {
        if (tid == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(tickets + t_list, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != chunk) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > (1u << 26)) {
                    lost = true;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        lost = __syncthreads_or(lost ? 1 : 0) != 0;
    }
    if (lost) {
        if (tid == 0) atomicOr(device_status, kStatusLostTicket);
    } else if (live_wave) {
        const bool slab_live = chunk > 0 || slab_live0 != 0;
        double* out = partial + (static_cast<int64_t>(ti) * (ti + 1) / 2 + tj) * (kSlab * kSlab);
        const double* un = unscale + static_cast<int64_t>(chunk) * rows_pad;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (!block_live(m, n)) continue;
                const int j = wc * 64 + 16 * n + (lane & 15);
                const double uj = un[static_cast<int64_t>(tj) * kSlab + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = row_in_slab + 16 * m + 4 * (lane >> 4) + e;
                    double v = static_cast<double>(acc2[m][n][e]);
                    v += static_cast<double>(acc[m][n][e]);
                    v *= un[static_cast<int64_t>(ti) * kSlab + i] * uj;   // powers of two: exact
                    if (slab_live) v += out[i * kSlab + j];
                    out[i * kSlab + j] = v;
                }
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(tickets + t_list, chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The deferred slab update: slab entry (i, j) of every tile of the launch += its chunks' sums in chunk order -- per chunk
// v = (double)level-1 sum (+ (double)unflushed chain), v *= 2^-shift_i 2^-shift_j, slab = v + slab: the fp64 operations of
// the in-kernel update, in its order.  One workgroup per (tile, slab, quarter of its rows); a thread owns four consecutive
// columns of a row and keeps kReduceRun chunks' loads in flight (the additions then run in chunk order: a first version with
// one dependent load per addition ran at 0.7 TB/s, 6 ms per launch).  Entries of blocks nobody computed are left alone.
constexpr int kReduceRun = 8;
__global__ __launch_bounds__(256) void chunk_reduce_kernel(const float* __restrict__ chunk_sums, const float* __restrict__ ragged_sums,
                                                           int n_tiles, int n_chunks, int ragged_chunk,
                                                           const int2* __restrict__ tile_order, const double* __restrict__ unscale,
                                                           int64_t rows_pad, double* __restrict__ partial, int slab_live0,
                                                           int t128, int n_blocks32) {
    typedef double f64x4 __attribute__((ext_vector_type(4)));
    const int t_list = blockIdx.x >> 1, sel = blockIdx.x & 1;
    const int2 tt = tile_order[t_list];
    const int ti = 2 * tt.x + sel, tj = tt.y;
    if (!(tj <= ti && ti < t128)) return;
    double* out = partial + (static_cast<int64_t>(ti) * (ti + 1) / 2 + tj) * (kSlab * kSlab);
    const int64_t slab_stride = static_cast<int64_t>(n_tiles) * 2 * (kSlab * kSlab);
    const float* in = chunk_sums + (static_cast<int64_t>(t_list) * 2 + sel) * (kSlab * kSlab);
    const float* rag = ragged_sums + (static_cast<int64_t>(t_list) * 2 + sel) * (kSlab * kSlab);
    // blockIdx.y: a quarter of the slab's rows; a thread: row i, columns j .. j + 3 (all in one 32 x 32 block)
    const int i = static_cast<int>(blockIdx.y) * 32 + (threadIdx.x >> 3), j = (threadIdx.x & 7) * 4;
    for (int jb = 0; jb < kSlab; jb += 32) {
        const int jj = jb + j;
        if (n_blocks32 >= 0) {       // the tile kernel's rule: block (i / 32, jj / 32) was computed iff ...
            const int rblk = ti * 4 + (i >> 5), cblk = tj * 4 + (jj >> 5);
            if (!(rblk < n_blocks32 && cblk <= rblk)) continue;
        }
        const int idx = i * kSlab + jj;
        f64x4 v = {0.0, 0.0, 0.0, 0.0};
        bool live = slab_live0 != 0;
        if (live) v = *reinterpret_cast<const f64x4*>(out + idx);
        for (int c0 = 0; c0 < n_chunks; c0 += kReduceRun) {
            f32x4 p[kReduceRun];
            double ui[kReduceRun];
            f64x4 uj[kReduceRun];
#pragma unroll
            for (int k = 0; k < kReduceRun; ++k) {
                const int c = c0 + k < n_chunks ? c0 + k : n_chunks - 1;
                p[k] = *reinterpret_cast<const f32x4*>(in + c * slab_stride + idx);
                const double* un = unscale + static_cast<int64_t>(c) * rows_pad;
                ui[k] = un[static_cast<int64_t>(ti) * kSlab + i];
                uj[k] = *reinterpret_cast<const f64x4*>(un + static_cast<int64_t>(tj) * kSlab + jj);
            }
#pragma unroll
            for (int k = 0; k < kReduceRun; ++k) {
                if (c0 + k >= n_chunks) break;
                f64x4 q = {static_cast<double>(p[k][0]), static_cast<double>(p[k][1]), static_cast<double>(p[k][2]),
                           static_cast<double>(p[k][3])};
                if (c0 + k == ragged_chunk) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(rag + idx);
                    q[0] += static_cast<double>(r[0]); q[1] += static_cast<double>(r[1]);
                    q[2] += static_cast<double>(r[2]); q[3] += static_cast<double>(r[3]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[e] *= ui[k] * uj[k][e];
                    v[e] = live ? q[e] + v[e] : q[e];
                }
                live = true;
            }
        }
        *reinterpret_cast<f64x4*>(out + idx) = v;
    }
}

int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : fallback;
}

}  // namespace

bool gram_planes_enabled() { return env_int("BYZ_GRAM_PLANES", 1) != 0; }

// Fills `slabs` (gram.hip's fp64 slab format: one 128 x 128 slab per lower-triangle tile ti (ti + 1) / 2 + tj) with the
// Gram of the n_rows logical rows G[row_index[r]]; with share_count > 1 only this share's tiles (owned[] says which).
// f16 = true: the f16x2 arithmetic, false: bf16x3 (bitwise gram.hip's split mode).
int launch_gram_planes(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                       double* slabs, int share_count, int share_index, uint8_t* owned_host, bool f16, hipStream_t stream) {
    const int64_t t128 = ceil_div(n_rows, kSlab);
    const int64_t t256 = ceil_div(t128, 2);
    const int64_t rows_pad = t256 * kWgRows;
    // BYZ_GRAM_MFMA=32: the f16x2 tile kernel on v_mfma_f32_32x32x16_f16 (rounds 2-5; the same-box A/B) instead of 16x16x32
    const bool shape16 = f16 && env_int("BYZ_GRAM_MFMA", 16) != 32;
    // The tile list.  BYZ_GRAM_ORDER=1 (the 16x16x32 kernel's default): bands of four 256-row blocks, inside a band column block
    // by column block -- ANY 32 consecutive tiles are (4 or 5) x (8 or 9) tiles on at most ~17 row blocks of 128, and the kernel
    // hands out runs of 32 consecutive (chunk, tile) units to the XCDs in turn.  BYZ_GRAM_ORDER=0 (rounds 2-5, and the other
    // kernels): 8 x 8 super-blocks of slabs = 4 x 8 workgroup tiles, a contiguous share of the list per XCD -- the rounds of an XCD
    // then straddle two chunks and two or three super-blocks: 106 GB through the fabric per 1M-column launch at N = 4000 if every
    // line were shared perfectly inside a round, against 70 GB for the bands (scripts/gram_order_footprint.py; EXPERIMENTS.md G8).
    const int order_mode = shape16 && env_int("BYZ_GRAM_ORDER", 1) != 0 ? 1 : 0;
    const int64_t order_key = static_cast<int64_t>(order_mode) * (1ll << 40) + share_count * 65536 + share_index;
    if (ctx->plane_order_T != t128 || ctx->plane_order_share != order_key) {
        ctx->plane_order_host.clear();
        int64_t position = 0;
        auto push = [&](int64_t bi, int64_t tj) {
            if (position++ % share_count != share_index) return;
            ctx->plane_order_host.push_back(static_cast<int32_t>(bi));
            ctx->plane_order_host.push_back(static_cast<int32_t>(tj));
        };
        if (order_mode == 1) {
            // (the bands cut into column panels of 16 / 24 / 32 / 40 slabs, so that at N = 10,000 -- a chunk of planes is 335 MB --
            // the rounds in flight stay inside the 256 MB Infinity Cache: measured, 91.7 -> 91.5 ms per launch: nothing; removed)
            for (int64_t b0 = 0; b0 < t256; b0 += 4)
                for (int64_t tj = 0; tj < t128 && tj <= 2 * (b0 + 3) + 1; ++tj)
                    for (int64_t bi = b0; bi < b0 + 4 && bi < t256; ++bi)
                        if (tj <= 2 * bi + 1) push(bi, tj);
        } else {
            const int64_t S = ceil_div(t128, 8);
            for (int64_t I = 0; I < S; ++I)
                for (int64_t J = 0; J <= I; ++J)
                    for (int64_t bi = I * 4; bi < I * 4 + 4 && bi < t256; ++bi)
                        for (int64_t tj = J * 8; tj < J * 8 + 8 && tj <= 2 * bi + 1 && tj < t128; ++tj) push(bi, tj);
        }
        BYZ_TRY(ctx->plane_order.ensure(ctx->plane_order_host.size() * sizeof(int32_t) + 16));
        BYZ_HIP(hipMemcpyAsync(ctx->plane_order.ptr, ctx->plane_order_host.data(),
                               ctx->plane_order_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BYZ_HIP(hipStreamSynchronize(stream));
        ctx->plane_order_T = t128;
        ctx->plane_order_share = order_key;
    }
    const int64_t n_tiles = static_cast<int64_t>(ctx->plane_order_host.size() / 2);
    if (owned_host != nullptr)
        for (int64_t t = 0; t < n_tiles; ++t) {
            const int64_t bi = ctx->plane_order_host[2 * t], tj = ctx->plane_order_host[2 * t + 1];
            for (int64_t ti = 2 * bi; ti < 2 * bi + 2 && ti < t128; ++ti)
                if (tj <= ti) owned_host[ti * (ti + 1) / 2 + tj] = 1;
        }
    if (n_tiles == 0) return BYZ_OK;

    // super-chunk: as many 8192-column chunks as the plane budget holds
    const int n_planes = f16 ? 2 : 3;
    size_t free_b = 0, total_b = 0;
    BYZ_HIP(hipMemGetInfo(&free_b, &total_b));
    int64_t budget = static_cast<int64_t>(env_int("BYZ_GRAM_PLANE_MB", 16384)) << 20;
    const int64_t have = static_cast<int64_t>(ctx->gram_planes.bytes);
    if (budget > have + static_cast<int64_t>(free_b / 2)) budget = have + static_cast<int64_t>(free_b / 2);
    int64_t chunks_per_sc = budget / (rows_pad * 2 * n_planes * kChunkCols);
    const int64_t chunks_total = ceil_div(n_cols, kChunkCols);
    if (chunks_per_sc > chunks_total) chunks_per_sc = chunks_total;
    if (chunks_per_sc < 1) {
        set_error("gram: no room for one 8192-column chunk of 16-bit planes (%lld rows)", (long long)rows_pad);
        return BYZ_E_HIP;
    }
    // even super-chunks: the same number of launches, the last one not a stub
    const int64_t n_sc = ceil_div(chunks_total, chunks_per_sc);
    chunks_per_sc = ceil_div(chunks_total, n_sc);
    const int64_t sc_cols = chunks_per_sc * kChunkCols;
    BYZ_TRY(ctx->gram_planes.ensure(static_cast<size_t>(rows_pad) * 2 * n_planes * sc_cols));
    BYZ_TRY(ctx->gram_tickets.ensure(static_cast<size_t>(n_tiles + 8) * sizeof(int)));
    double* unscale = nullptr;
    if (f16) {
        BYZ_TRY(ctx->plane_unscale.ensure(static_cast<size_t>(rows_pad) * chunks_per_sc * sizeof(double)));
        unscale = ctx->plane_unscale.as<double>();
    }
    int* tickets = ctx->gram_tickets.as<int>();
    u32x4* planes = ctx->gram_planes.as<u32x4>();
    typedef void (*kernel_t)(const u32x4*, int64_t, const double*, int64_t, double*, int, const int2*, int, int*, int, int,
                             int, int, int32_t*, float*, float*, int, int, int);
    // BYZ_GRAM_DEFER=0: round 4's in-kernel slab update (the same-box A/B and the bitwise comparison of the tests)
    bool defer = f16 && env_int("BYZ_GRAM_DEFER", 1) != 0;
    kernel_t kernel = !f16 ? &gram_planes_kernel<3, 4, 0>
                      : shape16 ? (defer ? &gram_planes16_kernel<6, true> : &gram_planes16_kernel<6, false>)
                                : (defer ? &gram_planes_kernel<2, 6, 0, 2, true> : &gram_planes_kernel<2, 6, 0>);
    int nbuf = f16 ? 6 : 4;
    const int threads = kThreads;
#ifdef BYZ_GRAM_DEBUG_VARIANTS
    // Timing experiments with WRONG results (the DBG bits of the kernel, times ten; scripts/gram_ab.py, EXPERIMENTS.md G1-G3).
    // Not compiled into the shipped library: build with -DBYZ_GRAM_DEBUG_VARIANTS to get them back.
    {
        const int variant = env_int("BYZ_GRAM_PLANES_VARIANT", 0);
        if (variant != 0) defer = false;
        if (f16) {
            kernel = variant == 10 ? &gram_planes_kernel<2, 6, 1> : variant == 20 ? &gram_planes_kernel<2, 6, 2>
                     : variant == 50 ? &gram_planes_kernel<2, 6, 5>      // no DMA, no barrier
                     : variant == 90 ? &gram_planes_kernel<2, 6, 9>      // no DMA, no LDS reads
                     : variant == 130 ? &gram_planes_kernel<2, 6, 13>    // the MFMAs, the loop and the slab update only
                     : variant == 160 ? &gram_planes_kernel<2, 6, 16>    // everything but the slab update
                     : variant == 340 ? &gram_planes_kernel<2, 6, 34>    // DMA only, every workgroup the same tile
                     : variant == 320 ? &gram_planes_kernel<2, 6, 32>    // everything, every workgroup the same tile
                                      : kernel;
        } else {
            kernel = variant == 10 ? &gram_planes_kernel<3, 4, 1> : variant == 20 ? &gram_planes_kernel<3, 4, 2> : kernel;
        }
    }
#endif
    float* chunk_sums = nullptr;
    float* ragged_sums = nullptr;
    if (defer) {
        // fp32 level-1 sums of every (chunk, tile, slab) of a launch + one more slab pair per tile for an unflushed chain
        // (4.3 GB at configs[3], 10 GB at configs[4]'s slice).  Where that does not fit beside the matrix the update stays
        // inside the tile kernel: slower, the same bits.
        const size_t per_chunk = static_cast<size_t>(n_tiles) * 2 * kSlab * kSlab * sizeof(float);
        const size_t want = per_chunk * static_cast<size_t>(chunks_per_sc + 1);
        size_t free_now = 0, total_now = 0;
        BYZ_HIP(hipMemGetInfo(&free_now, &total_now));
        if (want > ctx->gram_chunk_sums.bytes && want - ctx->gram_chunk_sums.bytes > free_now / 2) {
            defer = false;
            kernel = shape16 ? &gram_planes16_kernel<6, false> : &gram_planes_kernel<2, 6, 0>;
        }
    }
    if (defer) {
        const size_t per_chunk = static_cast<size_t>(n_tiles) * 2 * kSlab * kSlab * sizeof(float);
        BYZ_TRY(ctx->gram_chunk_sums.ensure(per_chunk * static_cast<size_t>(chunks_per_sc + 1)));
        chunk_sums = ctx->gram_chunk_sums.as<float>();
        ragged_sums = chunk_sums + (per_chunk / sizeof(float)) * static_cast<size_t>(chunks_per_sc);
    }
    const size_t lds_bytes = static_cast<size_t>(nbuf) * kRowBlocks * n_planes * kFragBytes;
    BYZ_HIP(allow_dynamic_lds(ctx, reinterpret_cast<const void*>(kernel), static_cast<int>(lds_bytes)));
    const int round_size = env_int("BYZ_GRAM_ROUND", ctx->num_cus / 8);   // one workgroup per CU
    const int n_blocks32 = env_int("BYZ_GRAM_BLOCK_SKIP", 1) != 0 ? static_cast<int>(ceil_div(n_rows, 32)) : -1;
    const int64_t per_xcd = ceil_div(n_tiles, 8);
    for (int64_t sc = 0; sc < n_sc; ++sc) {
        const int64_t k0 = sc * sc_cols;
        const int64_t cols = n_cols - k0 < sc_cols ? n_cols - k0 : sc_cols;
        const int64_t n_steps = ceil_div(cols, 32) * 2;   // whole 32-column stages, as gram.hip counts them (zero-filled)
        const int64_t n_chunks = ceil_div(n_steps, kChunkSteps);
        {
            KernelTimer t(ctx, BYZ_K_PLANE_SPLIT, stream);
            if (f16) {
                const dim3 grid(static_cast<unsigned>(n_chunks), static_cast<unsigned>(rows_pad / 32));
                if (env_int("BYZ_GRAM_SPLIT_TWO_PASS", 0) != 0) {   // round 2's kernel for everything (the comparison)
                    plane_split_f16_kernel<<<grid, 256, 0, stream>>>(G, n_rows, n_cols, ld, row_index, k0, n_steps, planes,
                                                                     unscale, rows_pad, nullptr);
                } else {
                    // one pass with a sampled scale; the blocks where it did not hold are listed and redone exactly
                    const int64_t pairs = n_chunks * (rows_pad / 32);
                    BYZ_TRY(ctx->split_redo.ensure(static_cast<size_t>(1 + 2 * pairs) * sizeof(int32_t)));
                    int32_t* redo = ctx->split_redo.as<int32_t>();
                    BYZ_HIP(hipMemsetAsync(redo, 0, sizeof(int32_t), stream));
                    // (non-temporal loads of G here: measured in round 6, 6.38 ms per launch either way -- EXPERIMENTS.md G6)
                    plane_split_f16_stream_kernel<<<grid, 256, 0, stream>>>(G, n_rows, n_cols, ld, row_index, k0, n_steps, planes,
                                                                            unscale, rows_pad, redo);
                    BYZ_TRY(check_launch("plane_split_f16_stream_kernel"));
                    // (sized for a few thousand listed blocks; the surplus workgroups leave at once, and a longer list --
                    // pathological data -- is walked in a grid-stride loop: no host read-back of the count.)
                    const int64_t fix = pairs < 4096 ? pairs : 4096;
                    plane_split_f16_kernel<<<static_cast<unsigned>(fix), 256, 0, stream>>>(G, n_rows, n_cols, ld, row_index, k0,
                                                                                           n_steps, planes, unscale, rows_pad, redo);
                }
            } else {
                const dim3 grid(static_cast<unsigned>(ceil_div(n_steps, kSplitCols / 16)), static_cast<unsigned>(rows_pad / 32));
                plane_split_bf16_kernel<<<grid, 256, 0, stream>>>(G, n_rows, n_cols, ld, row_index, k0, n_steps, planes);
            }
            BYZ_TRY(check_launch("plane_split_kernel"));
        }
        {
            // chunks per workgroup (DEFER only): BYZ_GRAM_KSPAN as given, else kDefaultSpan -- but never so many that the launch
            // has fewer than ~8 workgroups per CU (the tail of the last round would cost more than the turnover saves)
            int64_t kspan = 1;
            if (defer) {
                kspan = env_int("BYZ_GRAM_KSPAN", 0);
                if (kspan > n_chunks) kspan = n_chunks;   // (a span longer than the launch is the launch; keeps span x steps inside an int)
                if (kspan < 1) {
                    kspan = kDefaultSpan;
                    while (kspan > 1 && 8 * per_xcd * ceil_div(n_chunks, kspan) < static_cast<int64_t>(ctx->num_cus) * 8) kspan /= 2;
                }
            }
            // (order_mode 1: whole rounds of `map_round` units, dealt to the XCDs in turn)
            const int run = order_mode == 1 ? (round_size > 0 ? round_size : ctx->num_cus / 8) : 0;
            const int64_t n_units = n_tiles * ceil_div(n_chunks, kspan);
            int64_t rounds_per_xcd = run > 0 ? ceil_div(ceil_div(n_units, run), 8) : 0;
            // BYZ_GRAM_SPARE: spare rounds per XCD, in percent (default 10).  Every XCD is launched the same number of rounds; with
            // claims and a few rounds to spare a faster XCD takes more runs and the surplus rounds of the others find nothing left
            // and leave at once (same box: 37.57 -> 37.30 ms per launch at N = 4000, 92.1 -> 90.9 at N = 10,000; 100 %: 37.87)
            if (run > 0) rounds_per_xcd += rounds_per_xcd * env_int("BYZ_GRAM_SPARE", 10) / 100;
            const int64_t grid = run > 0 ? 8 * run * rounds_per_xcd : 8 * per_xcd * ceil_div(n_chunks, kspan);
            // BYZ_GRAM_CLAIM=0: the XCDs take the runs in turn; default: every round of an XCD CLAIMS the next run from one counter.
            // The XCDs of one chip do not run at one speed: dealt in turn, the launch waits for its slowest XCD with the others
            // idle (same box, N = 4000: 41.4 -> 39.1 ms per launch; N = 10,000: 100.8 -> 98.6; profiles/r06w_*)
            const bool claim = run > 0 && env_int("BYZ_GRAM_CLAIM", 1) != 0;
            const int map_round = claim ? -run : run;
            // tickets: [n_tiles] chunk order of a tile (in-kernel update), [8] workgroups done per XCD, [1] runs claimed,
            // [8][rounds_per_xcd] the run each round of an XCD claimed (+ 1; 0: not yet)
            const size_t ticket_ints = static_cast<size_t>(n_tiles + 9 + 8 * rounds_per_xcd);
            BYZ_TRY(ctx->gram_tickets.ensure(ticket_ints * sizeof(int)));
            tickets = ctx->gram_tickets.as<int>();
            BYZ_HIP(hipMemsetAsync(tickets, 0, ticket_ints * sizeof(int), stream));
            KernelTimer t(ctx, BYZ_K_GRAM, stream);
            if (grid > 0x7fffffff) {
                set_error("gram: grid too large");
                return BYZ_E_UNSUPPORTED;
            }
            kernel<<<static_cast<unsigned>(grid), threads, lds_bytes, stream>>>(
                planes, n_steps, unscale, rows_pad, slabs, static_cast<int>(n_tiles), ctx->plane_order.as<int2>(),
                static_cast<int>(n_chunks), tickets, round_size, static_cast<int>(t128), sc > 0 ? 1 : 0,
                n_blocks32, device_status_word(ctx), chunk_sums, ragged_sums, static_cast<int>(kspan), env_int("BYZ_GRAM_PIN", 1),
                map_round);
            BYZ_TRY(check_launch("gram_planes_kernel"));
        }
        if (defer) {
            KernelTimer t(ctx, BYZ_K_GRAM_REDUCE, stream);
            const int ragged_chunk = (n_steps % kChunkSteps) % kFlushSteps != 0 ? static_cast<int>(n_chunks) - 1 : -1;
            chunk_reduce_kernel<<<dim3(static_cast<unsigned>(2 * n_tiles), 4), 256, 0, stream>>>(
                chunk_sums, ragged_sums, static_cast<int>(n_tiles), static_cast<int>(n_chunks), ragged_chunk,
                ctx->plane_order.as<int2>(), unscale, rows_pad, slabs, sc > 0 ? 1 : 0, static_cast<int>(t128), n_blocks32);
            BYZ_TRY(check_launch("chunk_reduce_kernel"));
        }
    }
    return BYZ_OK;
}

}  // namespace byz
