// Krum for the reference's own sizes, N <= 128 clients (defences.py:16-42): five short launches instead of fifteen.
//
// The general path (gram.hip + select.hip) is built for N in the thousands: at N = 100, D = 79,510 (BASELINE configs[1]) it
// spends 69 us per round in ~15 kernel launches and memsets, most of it waiting for the host to issue them (~3.5 us each),
// and its exact fp32-input MFMA Gram alone takes 17-30 us.  A kernel boundary costs ~1.5-1.9 us on this part and an
// in-kernel grid barrier 4-7 us (MI355X_MICROARCH.md, price list), so the phases below are separate launches wherever
// EVERY workgroup needs EVERY other workgroup's output, and one launch wherever one workgroup can carry on alone:
//
//   K1 small_gram_kernel     all N rows x a 128-column slice per step, one row of the slice per thread: global fp32 ->
//                            registers -> (row, slice) power-of-two scale -> two fp16 planes in LDS (row-major, 272-byte
//                            pitch: conflict-free ds_write_b64 and ds_read_b128) -> v_mfma_f32_32x32x16_f16, three per
//                            32 x 32 block and 16 columns (m h' + h m' + h h': gram_planes.hip's f16x2 arithmetic, 6e-8
//                            against fp64), lower-triangle blocks only, spread over the 8 waves so that no SIMD carries
//                            more than 3; the next two slices' loads are in flight meanwhile.  One fp32 slab (<= 10 blocks
//                            x 4 KiB) and one compact diagonal per workgroup.
//   K2 small_reduce_kernel   slabs -> fp64 Gram blocks, 64 entries per workgroup, fixed summation order; c_ii and c_jj by the
//                            same sums out of the compact diagonals, so d_ij = sqrt(max(0, c_ii + c_jj - 2 c_ij)) is formed
//                            here, on 160 CUs; two counters record pairs the identity cannot resolve and exact zeros.
//   K3 small_distance_kernel leaves at once while both counters are zero.  Otherwise ONE workgroup redoes the distances with
//                            identical rows folded and near-duplicate pairs listed exactly as gram.hip does (d^2 <
//                            (c_ii + c_jj) / 16), helper workgroups of the same launch re-evaluate the listed pairs on the
//                            difference itself (defences.py:20), and identical rows end up with bitwise identical distance
//                            rows (the exact ties the reference resolves by visit order).
//   K4 small_score_kernel    one wave per row: in-register bitonic sort of the row's distances (values only: equal
//                            values add up the same in any order), then the SEQUENTIAL fp32 sum of the first n - f of
//                            them, exactly as Python's sum() forms it (defences.py:33-34).
//   K5 small_pick_kernel     every workgroup finds the winner itself (visit order 1, 0, 2, ..., strict '<' against 1e20,
//                            defences.py:27-37) and copies its share of the winning row.
//
// Algorithmic traffic: 4 N D bytes read once (K1); everything else is O(N^2).  Bound: HBM (N / 4 flop per byte is below
// the machine balance of the 16-bit matrix pipe for every N <= 128).
#include "common.hpp"
#include "lane_exchange.hpp"

#include <cstdlib>
#include <type_traits>

namespace byz {
namespace {

constexpr int kThreads = 512;
constexpr int kMaxRows = 128;
constexpr int kSlice = 128;                         // columns per step of K1
constexpr int kSteps = kSlice / 16;                 // MFMA k-steps per slice
constexpr int kPitch = 2 * kSlice + 16;             // bytes per row and plane in LDS: 272 (17 x 16)
constexpr int kPlaneBytes = kMaxRows * kPitch;      // 34,816
constexpr int kGramLds = 2 * kPlaneBytes + kMaxRows * 4;   // two planes + the rows' shifts
constexpr int kBlockEntries = 32 * 32;
constexpr int kMaxBlocks = 10;                      // lower triangle of 4 x 4 blocks
constexpr int kDistPitch = kMaxRows + 1;            // floats; odd pitch: a column walk touches every bank
constexpr int kPairChunk = 8192;                    // columns per (pair, chunk) work item of the near-duplicate pass
constexpr double kNearEps = 1.0 / 16.0;             // gram.hip's threshold
constexpr float kKrumInit = 1e20f;                  // defences.py:27
constexpr int kStatusPairOverflow = 2;              // bits of the context's sticky status word (gram.hip)
constexpr int kStatusFalseTwin = 4;
constexpr int kStatusSmallTimeout = 8;
constexpr unsigned kSpinLimit = 1u << 18;   // ~0.1-0.3 s: the worker publishes within microseconds

// LDS of K3 (one dynamic array, carved by hand)
constexpr int kK3Gram = 0;                                           // fp64 Gram blocks
constexpr int kK3Dist = kMaxBlocks * kBlockEntries * 8;              // 81,920: float [128][129]
constexpr int kK3Misc = kK3Dist + kMaxRows * kDistPitch * 4;         // 147,968: rep[128], rep2[128], counters
constexpr int kK3Lds = kK3Misc + 2 * kMaxRows * 4 + 64;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int block_index(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
__device__ __forceinline__ int visit_position(int u) { return u == 0 ? 1 : (u == 1 ? 0 : u); }

// 4 consecutive floats of a row starting at column k (any 4-byte alignment), zero past n_cols.  Branch-free: every lane
// loads (a clamped address) and selects afterwards -- a branch around a load makes hipcc wait for each load in turn.
__device__ __forceinline__ f32x4 load4_guarded(const float* __restrict__ row, int64_t k, int64_t n_cols) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int64_t c = k + e < n_cols ? k + e : n_cols - 1;
        const float x = row[c];
        v[e] = k + e < n_cols ? x : 0.0f;
    }
    return v;
}

// ---- K1 ---------------------------------------------------------------------------------------------------------------
// Workgroup w takes slices w, w + grid, w + 2 grid, ...  Wave q loads rows 16 q .. 16 q + 15, two rows (2 x 512 bytes) per
// instruction: lane l holds columns 4 (l & 31) .. + 3 of row 16 q + 2 i + (l >> 5) in v[i].
//
// Every slice is loaded by the same eight unconditional 16-byte loads (hipcc counts outstanding loads statically: one
// conditional load in the pipeline and every wait becomes "everything"):
//   * the last, ragged slice reads the window [n_cols - 128, n_cols) instead and zeroes the columns the slice before it
//     already covered (the order of the columns inside a slice does not matter to a sum over them);
//   * past the last slice the prefetch reads one 16-byte word of the matrix over and over.
// TINY (n_cols < 128: one slice, one workgroup) is the exception: element-wise guarded loads.
// PER > 0: every workgroup consumes exactly PER slices, fully unrolled (slices past the last one contribute zeros): with no
// loop the compiler's load counting is exact and two slices really are in flight behind the one being multiplied; PER == 0
// is the loop form for long rows (its header waits for everything outstanding).
template <bool TINY, int PER>
__global__ __launch_bounds__(kThreads, 2) void small_gram_kernel(const float* __restrict__ G, int n_rows, int64_t n_cols,
                                                                 int64_t ld, int n_slices, float* __restrict__ slabs,
                                                                 float* __restrict__ diag_slabs, int32_t* __restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    int* shifts = reinterpret_cast<int*>(lds + 2 * kPlaneBytes);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, q4 = lane & 31;
    const int n_rb = (n_rows + 31) >> 5;                 // live 32-row blocks
    const int n_blocks = n_rb * (n_rb + 1) / 2;
    if (blockIdx.x == 0 && tid == 0) {   // what K2 of this call will report to K3 (K3 of the previous call has read its own)
        flags[0] = 0;
        flags[1] = 0;
    }

    // lower-triangle blocks of this wave: waves w and w + 4 share a SIMD, no SIMD carries more than three blocks
    int bi0 = 0, bj0 = 0, bi1 = 0, bj1 = 0, nb = 0;
    switch (wave) {
        case 0: bi0 = 0; bj0 = 0; bi1 = 1; bj1 = 0; nb = 2; break;
        case 1: bi0 = 1; bj0 = 1; nb = 1; break;
        case 2: bi0 = 2; bj0 = 0; bi1 = 2; bj1 = 1; nb = 2; break;
        case 3: bi0 = 2; bj0 = 2; nb = 1; break;
        case 4: bi0 = 3; bj0 = 0; nb = 1; break;
        case 5: bi0 = 3; bj0 = 1; bi1 = 3; bj1 = 2; nb = 2; break;
        case 6: bi0 = 3; bj0 = 3; nb = 1; break;
        default: nb = 0; break;
    }
    if (nb == 2 && bi1 >= n_rb) nb = 1;    // row blocks past the matrix
    if (nb >= 1 && bi0 >= n_rb) nb = 0;

    // loads: a thread owns ONE row of the slice -- lane l of wave q holds columns 16 k + 4 (l & 3) .. + 3, k = 0 .. 7, of row
    // 16 q + (l >> 2): 64 contiguous bytes per row and instruction (half a cache line; the next instruction takes the other
    // half), and the row's largest magnitude costs 31 local max + 2 DPP steps instead of 5 cross-lane steps per register.
    const int r_local = 16 * wave + (lane >> 2);
    const int chunk = lane & 3;
    const float* src;
    {
        int r = r_local;
        if (r > n_rows - 1) r = n_rows - 1;   // rows past the matrix: clamped copies, they land in entries nobody reads
        src = G + static_cast<int64_t>(r) * ld;
    }
    const int64_t last_k0 = n_cols - kSlice;              // window of the ragged slice (TINY: unused)
    auto load_slice = [&](f32x4 (&v)[8], int s) __attribute__((always_inline)) {
        if constexpr (TINY) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = load4_guarded(src, 16 * k + 4 * chunk, n_cols);
        } else {
            const bool live = s < n_slices;                // uniform
            int64_t k0 = static_cast<int64_t>(s) * kSlice;
            if (k0 > last_k0) k0 = last_k0;
            const float* ptr = live ? src + k0 + 4 * chunk : G;
            const int step = live ? 16 : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f32x4u*>(ptr + k * step);
        }
    };

    f32x16 acc[2], sum[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            acc[b][e] = 0.0f;
            sum[b][e] = 0.0f;
        }

    auto multiply = [&](auto nb_c) __attribute__((always_inline)) {
        constexpr int NB = decltype(nb_c)::value;
        const unsigned char* a0 = lds + (32 * bi0 + q4) * kPitch + half * 16;
        const unsigned char* b0 = lds + (32 * bj0 + q4) * kPitch + half * 16;
        const unsigned char* a1 = lds + (32 * bi1 + q4) * kPitch + half * 16;
        const unsigned char* b1 = lds + (32 * bj1 + q4) * kPitch + half * 16;
#pragma unroll
        for (int t = 0; t < kSteps; ++t) {
            // m h' + h m' + h h' per block (gram_planes.hip's order), the two blocks' MFMAs interleaved
            f16x8 ah, am, bh, bm, ch, cm, dh, dm;
            ah = *reinterpret_cast<const f16x8*>(a0 + t * 32);
            am = *reinterpret_cast<const f16x8*>(a0 + t * 32 + kPlaneBytes);
            bh = *reinterpret_cast<const f16x8*>(b0 + t * 32);
            bm = *reinterpret_cast<const f16x8*>(b0 + t * 32 + kPlaneBytes);
            if constexpr (NB == 2) {
                ch = *reinterpret_cast<const f16x8*>(a1 + t * 32);
                cm = *reinterpret_cast<const f16x8*>(a1 + t * 32 + kPlaneBytes);
                dh = *reinterpret_cast<const f16x8*>(b1 + t * 32);
                dm = *reinterpret_cast<const f16x8*>(b1 + t * 32 + kPlaneBytes);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc[0], 0, 0, 0);
            if constexpr (NB == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cm, dh, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc[0], 0, 0, 0);
            if constexpr (NB == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, dm, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[0], 0, 0, 0);
            if constexpr (NB == 2) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, dh, acc[1], 0, 0, 0);
        }
        // undo the rows' scales (powers of two: exact) and add the slice to the running sums
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int bi = b == 0 ? bi0 : bi1, bj = b == 0 ? bj0 : bj1;
            const int sj = shifts[32 * bj + q4];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int si = shifts[32 * bi + (e & 3) + 8 * (e >> 2) + 4 * half];
                sum[b][e] += __builtin_ldexpf(acc[b][e], -(si + sj));
                acc[b][e] = 0.0f;
            }
        }
    };

    // one slice: scale + split into LDS, prefetch into the freed registers, MFMAs, unscaled into `sum`
    // MASKED: the slice may be the ragged one (its window overlaps the slice before it) or lie past the end (all zeros)
    auto consume = [&](f32x4 (&v)[8], int s, int next_slice, auto prefetch_c, auto masked_c) __attribute__((always_inline)) {
        constexpr bool kPrefetch = decltype(prefetch_c)::value;
        constexpr bool kMasked = decltype(masked_c)::value;
        if constexpr (kMasked && !TINY) {
            // columns of the ragged slice's window that belong to the slice before it
            const int64_t k0 = static_cast<int64_t>(s) * kSlice;
            const int covered = k0 > last_k0 ? static_cast<int>(k0 - last_k0) : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[k][e] = (16 * k + 4 * chunk + e >= covered) ? v[k][e] : 0.0f;
        }
        // Largest magnitude of the row's slice.  No special case for inf / NaN input: they make the scaled planes inf / NaN
        // whatever the shift is, and the poison reaches the Gram entries of that row as it would in any arithmetic.
        float mx = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = __builtin_fmaxf(mx, __builtin_fabsf(v[k][e]));
        mx = __builtin_fmaxf(mx, lanes::lane_xor(mx, 1, lane));
        mx = __builtin_fmaxf(mx, lanes::lane_xor(mx, 2, lane));
        int shift = 14 + 127 - static_cast<int>((__float_as_uint(mx) >> 23) & 0xffu);   // mx 2^shift in [2^14, 2^15)
        shift = shift > 126 ? 126 : shift;     // zero and subnormal magnitudes: as far up as a float scale goes
        shift = shift < -126 ? -126 : shift;
        const float scale = __uint_as_float(static_cast<uint32_t>(shift + 127) << 23);
        if (chunk == 0) shifts[r_local] = shift;
        unsigned char* dst = lds + r_local * kPitch + chunk * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            f16x4 h, m;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = v[k][e] * scale;
                h[e] = static_cast<_Float16>(x);
                m[e] = static_cast<_Float16>(x - static_cast<float>(h[e]));   // x - h is exact in fp32
            }
            *reinterpret_cast<u32x2*>(dst + 32 * k) = __builtin_bit_cast(u32x2, h);
            *reinterpret_cast<u32x2*>(dst + 32 * k + kPlaneBytes) = __builtin_bit_cast(u32x2, m);
        }
        __syncthreads();
        if constexpr (!TINY && kPrefetch) load_slice(v, next_slice);   // lands while the MFMAs run (a dummy word past the last slice)
        if (nb == 2) multiply(std::integral_constant<int, 2>{});
        else if (nb == 1) multiply(std::integral_constant<int, 1>{});
        __syncthreads();   // the planes and shifts are rewritten by the next slice
    };

    f32x4 va[8], vb[8];
    const int g = gridDim.x;
    constexpr std::true_type yes{};
    constexpr std::false_type no{};
    if constexpr (TINY) {
        load_slice(va, 0);
        consume(va, 0, 1, no, no);
    } else if constexpr (PER > 0) {
        // the host sizes the grid so that (PER - 1) * grid < n_slices: only the last step can hold the ragged slice or none
        const int s = blockIdx.x;
        load_slice(va, s);
        if constexpr (PER > 1) load_slice(vb, s + g);
#define BYZ_STEP(c, buf)                                                                          \
    if constexpr (PER > c) {                                                                      \
        if constexpr (PER > c + 2) consume(buf, s + c * g, s + (c + 2) * g, yes, no);             \
        else if constexpr (PER > c + 1) consume(buf, s + c * g, 0, no, no);                       \
        else consume(buf, s + c * g, 0, no, yes);                                                 \
    }
        BYZ_STEP(0, va) BYZ_STEP(1, vb) BYZ_STEP(2, va) BYZ_STEP(3, vb) BYZ_STEP(4, va) BYZ_STEP(5, vb) BYZ_STEP(6, va) BYZ_STEP(7, vb)
#undef BYZ_STEP
    } else {
        // long rows: four slices per trip (the loop header waits for every outstanding load, the other three do not)
        int s = blockIdx.x;
        load_slice(va, s);
        load_slice(vb, s + g);
        while (true) {
            consume(va, s, s + 2 * g, yes, yes);
            if (s + g >= n_slices) break;
            consume(vb, s + g, s + 3 * g, yes, yes);
            if (s + 2 * g >= n_slices) break;
            consume(va, s + 2 * g, s + 4 * g, yes, yes);
            if (s + 3 * g >= n_slices) break;
            consume(vb, s + 3 * g, s + 5 * g, yes, yes);
            s += 4 * g;
            if (s >= n_slices) break;
        }
    }

    float* out = slabs + static_cast<int64_t>(blockIdx.x) * n_blocks * kBlockEntries;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b < nb) {
            const int bi = b == 0 ? bi0 : bi1, bj = b == 0 ? bj0 : bj1;
            float* blk = out + block_index(bi, bj) * kBlockEntries;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int il = (e & 3) + 8 * (e >> 2) + 4 * half;
                blk[il * 32 + q4] = sum[b][e];
                // the diagonal once more, compactly: K2 needs c_ii and c_jj next to every c_ij
                if (bi == bj && il == q4) diag_slabs[static_cast<int64_t>(blockIdx.x) * kMaxRows + 32 * bi + q4] = sum[b][e];
            }
        }
    }
}

// ---- K2 ---------------------------------------------------------------------------------------------------------------
// gram[entry] = sum over the slabs, fp64, fixed order: wave q adds slabs q, q + 8, ... on four independent chains, the
// eight waves' sums are combined as a fixed tree.  Workgroup = 64 consecutive entries of one block (256 bytes per slab).
// c_ii and c_jj come out of the compact diagonal slabs by the very same summation (bitwise the sums of the workgroups that
// own those entries), so that d_ij = sqrt(max(0, c_ii + c_jj - 2 c_ij)) is formed right here, on 160 CUs instead of one.
// flags[0] counts the pairs the Gram identity cannot resolve (d^2 < (c_ii + c_jj) / 16: near-duplicate or identical rows),
// flags[1] the exact zeros: while both stay 0 -- the normal case -- K3 has nothing to do.
__global__ __launch_bounds__(kThreads) void small_reduce_kernel(const float* __restrict__ slabs,
                                                                const float* __restrict__ diag_slabs, int n_slabs, int n_blocks,
                                                                int n, double* __restrict__ gram, float* __restrict__ dist,
                                                                int32_t* __restrict__ flags) {
    __shared__ double part[3][8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int entry = static_cast<int>(blockIdx.x) * 64 + lane;
    const int blk = entry >> 10, local = entry & 1023;
    const int bi = blk >= 6 ? 3 : (blk >= 3 ? 2 : (blk >= 1 ? 1 : 0));
    const int bj = blk - bi * (bi + 1) / 2;
    const int il = local >> 5, jl = local & 31;
    const int i = 32 * bi + il, j = 32 * bj + jl;
    const int64_t slab = static_cast<int64_t>(n_blocks) * kBlockEntries;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    int sl = wave;
    for (; sl + 24 < n_slabs; sl += 32) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] += static_cast<double>(slabs[(sl + 8 * c) * slab + entry]);
            a[c] += static_cast<double>(diag_slabs[(sl + 8 * c) * kMaxRows + i]);
            b[c] += static_cast<double>(diag_slabs[(sl + 8 * c) * kMaxRows + j]);
        }
    }
    for (; sl < n_slabs; sl += 8) {
        s[0] += static_cast<double>(slabs[sl * slab + entry]);
        a[0] += static_cast<double>(diag_slabs[sl * kMaxRows + i]);
        b[0] += static_cast<double>(diag_slabs[sl * kMaxRows + j]);
    }
    part[0][wave][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    part[1][wave][lane] = (a[0] + a[1]) + (a[2] + a[3]);
    part[2][wave][lane] = (b[0] + b[1]) + (b[2] + b[3]);
    __syncthreads();
    if (wave != 0) return;
    double tot[3];
#pragma unroll
    for (int v = 0; v < 3; ++v)
        tot[v] = ((part[v][0][lane] + part[v][1][lane]) + (part[v][2][lane] + part[v][3][lane])) +
                 ((part[v][4][lane] + part[v][5][lane]) + (part[v][6][lane] + part[v][7][lane]));
    const double cij = tot[0], cii = tot[1], cjj = tot[2];
    gram[entry] = cij;
    if (i >= n || j >= n || j > i) return;   // padding rows; the upper half of a diagonal block (its mirror image is written)
    if (i == j) {
        dist[static_cast<int64_t>(i) * n + i] = __builtin_inff();   // the reference keeps no self-distance (defences.py:18-20)
        return;
    }
    const double d2 = cii + cjj - 2.0 * cij;
    const float d = static_cast<float>(sqrt(d2 < 0.0 ? 0.0 : d2));   // NaN (poisoned input) stays NaN
    dist[static_cast<int64_t>(i) * n + j] = d;
    dist[static_cast<int64_t>(j) * n + i] = d;
    if (d2 < kNearEps * (cii + cjj)) atomicAdd(flags + 0, 1);
    if (d == 0.0f) atomicAdd(flags + 1, 1);
}

// ---- K3 ---------------------------------------------------------------------------------------------------------------
struct DistanceArgs {
    const double* gram;        // K2's blocks
    const float* G;
    int n_rows;
    int64_t n_cols, ld;
    float* dist;               // n x n, pitch n
    int32_t* rep;              // n: scratch in global memory (debugging aid only)
    int2* pairs;               // near-duplicate pairs
    int pair_capacity;
    double* pair_partial;      // (pair, chunk) sums of squared differences
    int64_t item_capacity;
    int32_t* sync;             // [0] flag (epoch) [1] published pair count [2] arrivals [4], [5] K2's findings
    int32_t epoch;
    int32_t* status;
};

// c_ij of the lower triangle (i >= j) out of the fp64 blocks in LDS
__device__ __forceinline__ double gram_at(const double* gl, int i, int j) {
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    return gl[block_index(hi >> 5, lo >> 5) * kBlockEntries + (hi & 31) * 32 + (lo & 31)];
}

__device__ __forceinline__ unsigned long long bits_of(double v) { return static_cast<unsigned long long>(__double_as_longlong(v)); }

// sum over one column chunk of (g_a - g_b)^2: the difference in fp32 as the reference forms it (defences.py:20), squares
// and sums in fp64; a fixed partition over the 512 threads and a fixed tree, so the result does not depend on who runs it
__device__ double pair_chunk_sum(const DistanceArgs& p, int pair, int chunk, double* red) {
    const int2 pr = p.pairs[pair];
    const float* a = p.G + static_cast<int64_t>(pr.x) * p.ld;
    const float* b = p.G + static_cast<int64_t>(pr.y) * p.ld;
    const int64_t k0 = static_cast<int64_t>(chunk) * kPairChunk;
    const int64_t k1 = k0 + kPairChunk < p.n_cols ? k0 + kPairChunk : p.n_cols;
    const int64_t kv = k0 + ((k1 - k0) & ~static_cast<int64_t>(3));
    double acc = 0.0;
    for (int64_t k = k0 + 4 * threadIdx.x; k < kv; k += 4 * kThreads) {
        const f32x4 x = *reinterpret_cast<const f32x4u*>(a + k), y = *reinterpret_cast<const f32x4u*>(b + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double df = static_cast<double>(__fsub_rn(x[e], y[e]));
            acc = fma(df, df, acc);
        }
    }
    for (int64_t k = kv + threadIdx.x; k < k1; k += kThreads) {
        const double df = static_cast<double>(__fsub_rn(a[k], b[k]));
        acc = fma(df, df, acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = kThreads / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    const double total = red[0];
    __syncthreads();
    return total;
}

// A helper of the near-duplicate pass: waits for the worker's pair count, leaves when it is zero, otherwise takes
// (pair, chunk) items and reports back through one counter.  Every wait is bounded; a time-out sets the status word.
__device__ __forceinline__ void distance_helper(const DistanceArgs& p, unsigned char* lds) {
    double* gl = reinterpret_cast<double*>(lds + kK3Gram);
    float* dl = reinterpret_cast<float*>(lds + kK3Dist);
    int* rep = reinterpret_cast<int*>(lds + kK3Misc);
    int* rep2 = rep + kMaxRows;
    int* words = rep2 + kMaxRows;          // [0] pair counter [1] broadcast slot
    double* red = reinterpret_cast<double*>(lds + kK3Gram);   // helpers only (they never hold a Gram)
    const int tid = threadIdx.x;
    const int n = p.n_rows;
    const int n_chunks = static_cast<int>((p.n_cols + kPairChunk - 1) / kPairChunk);
    const int helpers = static_cast<int>(gridDim.x) - 1;
    (void)gl; (void)dl; (void)rep; (void)rep2; (void)words; (void)red; (void)n; (void)n_chunks; (void)helpers;
    // ---- helper: wait for the worker's pair count
    if (tid == 0) {
        unsigned spins = 0;
        int seen = 0;
        while (__hip_atomic_load(p.sync + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > kSpinLimit) {
                seen = -2;
                break;
            }
        }
        if (seen == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            seen = __hip_atomic_load(p.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        words[1] = seen;
    }
    __syncthreads();
    const int count = words[1];
    __syncthreads();
    if (count == -2 && tid == 0) atomicOr(p.status, kStatusSmallTimeout);
    if (count <= 0) return;
    const int64_t items = static_cast<int64_t>(count) * n_chunks;
    for (int64_t w = blockIdx.x - 1; w < items; w += helpers) {
        const double v = pair_chunk_sum(p, static_cast<int>(w / n_chunks), static_cast<int>(w % n_chunks), red);
        if (tid == 0) p.pair_partial[w] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(p.sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
}

// The worker (workgroup 0): representatives, distances, the pair list; after the helpers' sums the canonical distance rows.
__device__ __forceinline__ void distance_worker(const DistanceArgs& p, unsigned char* lds) {
    double* gl = reinterpret_cast<double*>(lds + kK3Gram);
    float* dl = reinterpret_cast<float*>(lds + kK3Dist);
    int* rep = reinterpret_cast<int*>(lds + kK3Misc);
    int* rep2 = rep + kMaxRows;
    int* words = rep2 + kMaxRows;          // [0] pair counter [1] broadcast slot
    double* red = reinterpret_cast<double*>(lds + kK3Gram);   // helpers only (they never hold a Gram)
    const int tid = threadIdx.x;
    const int n = p.n_rows;
    const int n_chunks = static_cast<int>((p.n_cols + kPairChunk - 1) / kPairChunk);
    const int helpers = static_cast<int>(gridDim.x) - 1;
    (void)gl; (void)dl; (void)rep; (void)rep2; (void)words; (void)red; (void)n; (void)n_chunks; (void)helpers;
    // ---- worker
    const int lane = tid & 63, wave = tid >> 6;
    const int n_rb = (n + 31) >> 5;
    const int n_entries = n_rb * (n_rb + 1) / 2 * kBlockEntries;
    {   // the Gram blocks into LDS: every load issued before the first use (a loop of dependent round trips otherwise)
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        constexpr int kPer = kMaxBlocks * kBlockEntries / 2 / kThreads;   // 10 16-byte loads per thread
        const f64x2* src = reinterpret_cast<const f64x2*>(p.gram);
        const int n_vec = n_entries / 2;
        f64x2 tmp[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int e = tid + k * kThreads;
            tmp[k] = src[e < n_vec ? e : 0];
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int e = tid + k * kThreads;
            if (e < n_vec) reinterpret_cast<f64x2*>(gl)[e] = tmp[k];
        }
    }
    if (tid == 0) words[0] = 0;
    __syncthreads();
    // rep[i] = the first j < i whose Gram entries are bitwise those of i (c_ij == c_ii == c_jj): identical rows nominate
    // each other this way in any arithmetic; the proof is the pair (i, rep[i]) on the list below.  One wave per row, the
    // lanes look at two candidates each.
    for (int i = wave; i < n; i += kThreads / 64) {
        const unsigned long long cii = bits_of(gram_at(gl, i, i));
        const int j0 = lane, j1 = lane + 64;
        const bool h0 = j0 < i && bits_of(gram_at(gl, i, j0)) == cii && bits_of(gram_at(gl, j0, j0)) == cii;
        const bool h1 = j1 < i && bits_of(gram_at(gl, i, j1 < n ? j1 : 0)) == cii && bits_of(gram_at(gl, j1 < n ? j1 : 0, j1 < n ? j1 : 0)) == cii;
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
        const int best = m0 ? __builtin_ctzll(m0) : (m1 ? 64 + __builtin_ctzll(m1) : i);
        if (lane == 0) {
            rep[i] = best;
            p.rep[i] = best;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int i = idx / n, j = idx - i * n;
        float d;
        if (i == j) {
            d = __builtin_inff();   // the reference keeps no self-distance (defences.py:18-20)
        } else {
            const double cii = gram_at(gl, i, i), cjj = gram_at(gl, j, j);
            const double d2 = cii + cjj - 2.0 * gram_at(gl, i, j);
            d = static_cast<float>(sqrt(d2 < 0.0 ? 0.0 : d2));   // NaN (poisoned input) stays NaN
            if (i > j && d2 < kNearEps * (cii + cjj)) {
                const int ri = rep[i], rj = rep[j];
                // representatives pair with each other; a folded row only with its representative (the proof of identity)
                if ((ri == i && rj == j) || ri == j) {
                    const int slot = atomicAdd(&words[0], 1);
                    if (slot < p.pair_capacity) p.pairs[slot] = make_int2(i, j);
                }
            }
        }
        dl[i * kDistPitch + j] = d;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int count = words[0];
    const bool overflow = count > p.pair_capacity || static_cast<int64_t>(count) * n_chunks > p.item_capacity;
    if (overflow) count = -1;
    if (tid == 0) {
        // publish the pair list and its length (zero lets the helpers go)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.sync + 1, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.sync + 0, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (overflow) atomicOr(p.status, kStatusPairOverflow);
    }
    if (overflow) return;   // the caller gets an error, not a half-patched matrix

    if (count > 0) {
        const int64_t items = static_cast<int64_t>(count) * n_chunks;
        bool timed_out = false;
        if (helpers > 0) {
            int expected = helpers;   // every helper reports once
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(p.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != expected) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > kSpinLimit) {
                        timed_out = true;
                        break;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(p.sync + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next call
            }
            timed_out = __syncthreads_or(timed_out ? 1 : 0) != 0;
        } else {
            // no helpers (a one-workgroup launch): the worker does the items itself; the Gram in LDS is no longer needed
            for (int64_t w = 0; w < items; ++w) {
                const double v = pair_chunk_sum(p, static_cast<int>(w / n_chunks), static_cast<int>(w % n_chunks), red);
                if (tid == 0) p.pair_partial[w] = v;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (timed_out) {
            if (tid == 0) atomicOr(p.status, kStatusSmallTimeout);
            return;
        }
        for (int q = tid; q < count; q += kThreads) {
            double sq = 0.0;
            for (int c = 0; c < n_chunks; ++c) sq += p.pair_partial[static_cast<int64_t>(q) * n_chunks + c];   // chunk order
            const int2 pr = p.pairs[q];
            // a row folded into pr.y whose difference from it is not zero after all: its other pairs were never listed
            if (rep[pr.x] == pr.y && sq != 0.0) atomicOr(p.status, kStatusFalseTwin);
            const float d = static_cast<float>(sqrt(sq));
            dl[pr.x * kDistPitch + pr.y] = d;
            dl[pr.y * kDistPitch + pr.x] = d;
        }
        __syncthreads();
    }

    // Identical rows must end up with bitwise identical distance rows (the reference resolves their exactly tied scores by
    // visit order): rep2[i] = the smallest j with d_ij == 0, chains followed to their root, every member of a group takes
    // the group's first row (gram.hip: canonicalise_duplicates).
    bool any_twin = false;
    for (int i = wave; i < n; i += kThreads / 64) {
        const int j0 = lane, j1 = lane + 64;
        const bool h0 = j0 < i && dl[i * kDistPitch + j0] == 0.0f;
        const bool h1 = j1 < i && dl[i * kDistPitch + j1] == 0.0f;
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
        const int best = m0 ? __builtin_ctzll(m0) : (m1 ? 64 + __builtin_ctzll(m1) : i);
        if (lane == 0) rep2[i] = best;
        any_twin = any_twin || best != i;
    }
    any_twin = __syncthreads_or(any_twin ? 1 : 0) != 0;
    if (any_twin) {
        for (int round = 0; round < 8; ++round) {   // rep2[i] < i along a chain: pointer jumping, log2(128) rounds at most
            int r = 0, rr = 0;
            if (tid < n) {
                r = rep2[tid];
                rr = rep2[r];
            }
            __syncthreads();
            if (tid < n && rr != r) rep2[tid] = rr;
            if (!__syncthreads_or(tid < n && rr != r ? 1 : 0)) break;
        }
    }
    for (int idx = tid; idx < n * n; idx += kThreads) {
        const int i = idx / n, j = idx - i * n;
        float d = dl[i * kDistPitch + j];
        if (i != j) {
            const int ri = rep2[i], rj = rep2[j];
            // reads touch only (root, root) entries, which nobody rewrites
            if (!(ri == i && rj == j)) d = ri == rj ? 0.0f : dl[ri * kDistPitch + rj];
        }
        p.dist[idx] = d;
    }
}

// Workgroup 0 is the worker, the others are helpers for the near-duplicate pass.
__global__ __launch_bounds__(kThreads) void small_distance_kernel(DistanceArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    // K2 has already written the distances; unless it met a pair the Gram identity cannot resolve (near-duplicate rows) or an
    // exact zero (identical rows), there is nothing left to do here.  Uniform over the grid: nobody waits for anybody.
    if (p.sync[4] == 0 && p.sync[5] == 0) return;
    if (blockIdx.x != 0) distance_helper(p, lds);
    else distance_worker(p, lds);
}

// ---- K4 ---------------------------------------------------------------------------------------------------------------
// One wave per row: the row's n - 1 distances (+inf in the self slot and past n) sorted ascending in registers, two per
// lane (index i = r + 2 lane), spilled to LDS in order, and lane 0 adds the first prefix_len of them left to right in fp32.
__global__ __launch_bounds__(256) void small_score_kernel(const float* __restrict__ dist, int n, int prefix_len,
                                                          float* __restrict__ scores) {
    __shared__ float sorted[4][kMaxRows];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int u = blockIdx.x * 4 + wave;
    if (u >= n) return;   // whole waves leave; nothing below synchronises the workgroup
    float x[1][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c = r + 2 * lane;
        x[0][r] = (c < n && c != u) ? dist[static_cast<int64_t>(u) * n + c] : __builtin_inff();
    }
    lanes::wave_bitonic_sort<2, 1>(x, lane);
    sorted[wave][2 * lane] = x[0][0];
    sorted[wave][2 * lane + 1] = x[0][1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
        float s = 0.0f;
        for (int r = 0; r < prefix_len; ++r) s = __fadd_rn(s, sorted[wave][r]);
        scores[u] = s;
    }
}

// ---- K5 ---------------------------------------------------------------------------------------------------------------

// Every workgroup repeats the (128-candidate) argmin, so that nobody waits for anybody: candidates in the reference's visit
// order 1, 0, 2, 3, ..., strict '<' against a running minimum that starts at 1e20 (no score below it: index -1).
__global__ __launch_bounds__(256) void small_pick_kernel(const float* __restrict__ scores, int n, const float* __restrict__ G,
                                                         int64_t n_cols, int64_t ld, int32_t* __restrict__ winner,
                                                         float* __restrict__ out_row) {
    __shared__ int chosen;
    const int tid = threadIdx.x;
    if (tid < 64) {
        float best = kKrumInit;
        int pos = 0x7fffffff, row = -1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int u = tid + 64 * r;
            if (u < n) {
                const float s = scores[u];
                const int vp = visit_position(u);
                if (s < kKrumInit && (s < best || (s == best && vp < pos))) {   // false for NaN, as in the reference
                    best = s;
                    pos = vp;
                    row = u;
                }
            }
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) {
            const float ob = __shfl_xor(best, m, 64);
            const int op = __shfl_xor(pos, m, 64);
            const int orow = __shfl_xor(row, m, 64);
            if (op != 0x7fffffff && (pos == 0x7fffffff || ob < best || (ob == best && op < pos))) {
                best = ob;
                pos = op;
                row = orow;
            }
        }
        if (tid == 0) {
            // a single row has an empty distance dict in the reference: nothing is visited, the index stays -1
            chosen = n < 2 ? -1 : row;
            if (blockIdx.x == 0) *winner = chosen;
        }
    }
    __syncthreads();
    if (out_row == nullptr) return;
    int64_t r = chosen;
    if (r < 0) r += n;   // numpy's G[-1]: the reference returns the last row when nothing won
    const float* src = G + r * ld;
    const int64_t per = (n_cols + gridDim.x - 1) / gridDim.x;
    const int64_t k0 = per * blockIdx.x;
    const int64_t k1 = k0 + per < n_cols ? k0 + per : n_cols;
    for (int64_t k = k0 + tid; k < k1; k += 256) out_row[k] = src[k];
}

// (K3 + K4 + K5 in ONE launch -- `small_tail_kernel`, written at the end of round 2 -- was measured in round 3 and removed:
// 30.5 against 30.4 us per round at D = 79,510.  The kernel trace (profiles/r03o_*) says why: the merged kernel takes 8.9 us
// where the three it replaces take 4.5 + 4.7 + 4.6, and the trace shows 10 us between its launches.  What a round costs
// on the GPU is K1 (16.0 us: 31.8 MB at 2.0 TB/s) plus four trivial dependent kernels at ~4.5 us EACH -- the price of a
// kernel boundary with its cache write-back on this part, not their work.)

int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : fallback;
}

}  // namespace

// BYZ_KRUM_SMALL: 0 = the general path, 1 (default) = this file.
bool krum_small_enabled() { return env_int("BYZ_KRUM_SMALL", 1) != 0; }

bool krum_small_applies(int64_t n_rows, int64_t n_cols) {
    // Default limit: 2^18 columns = up to eight unrolled 128-column slices for each of 256 workgroups (PER = 1 .. 8; the
    // reference's Cifar10Net, D = 117,706, is PER = 4).  All eight forms and the loop form behind them (up to 2^20 columns
    // with BYZ_KRUM_SMALL_MAX_COLS) are covered by tests/test_gpu_scale.py and ran green on the driver's box in round 2.
    int64_t max_cols = env_int("BYZ_KRUM_SMALL_MAX_COLS", 8 * 256 * kSlice);
    if (max_cols > (static_cast<int64_t>(1) << 20)) max_cols = static_cast<int64_t>(1) << 20;
    return krum_small_enabled() && n_rows >= 2 && n_rows <= kMaxRows && n_cols <= max_cols;
}

// byz_ctx_reserve's share: everything the N <= 128 path allocates, so that its first call allocates nothing
int reserve_small_workspaces(byz_ctx* ctx) {
    const size_t slab_floats = static_cast<size_t>(ctx->num_cus) * kMaxBlocks * kBlockEntries;
    BYZ_TRY(ctx->gram_partials.ensure((slab_floats + static_cast<size_t>(ctx->num_cus) * kMaxRows) * sizeof(float)));
    BYZ_TRY(ctx->gram.ensure(static_cast<size_t>(kMaxBlocks) * kBlockEntries * sizeof(double)));
    BYZ_TRY(ctx->gram_rep.ensure(static_cast<size_t>(kMaxRows) * sizeof(int32_t)));
    BYZ_TRY(ctx->near_pairs.ensure(static_cast<size_t>(kMaxRows * (kMaxRows - 1) / 2) * sizeof(int2)));
    BYZ_TRY(ctx->near_partial.ensure((static_cast<size_t>(1) << 20) * sizeof(double)));
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(kMaxRows) * sizeof(float)));
    if (ctx->small_sync.ptr == nullptr) {
        BYZ_TRY(ctx->small_sync.ensure(64));
        BYZ_HIP(hipMemset(ctx->small_sync.ptr, 0, 64));
    }
    return BYZ_OK;
}

// dist (n x n fp32, pitch n) of the n_rows x n_cols matrix G: K1..K3
static int small_distances(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* dist,
                           hipStream_t stream) {
    BYZ_REQUIRE(G && dist && n_rows >= 1 && n_rows <= kMaxRows && n_cols > 0 && ld >= n_cols,
                "small distances: bad shape %lld x %lld ld %lld", (long long)n_rows, (long long)n_cols, (long long)ld);
    const int n = static_cast<int>(n_rows);
    const int n_rb = (n + 31) / 32;
    const int n_blocks = n_rb * (n_rb + 1) / 2;
    const int64_t n_slices = ceil_div(n_cols, kSlice);
    // one workgroup per CU at most; the grid is sized so that everybody gets the same number of slices (+- 1)
    const int grid = static_cast<int>(ceil_div(n_slices, ceil_div(n_slices, ctx->num_cus)));
    const int64_t per = ceil_div(n_slices, grid);   // (per - 1) * grid < n_slices: only a workgroup's last slice can be ragged or missing
    // one slab of blocks per workgroup, then one compact diagonal per workgroup
    const size_t slab_floats = static_cast<size_t>(grid) * n_blocks * kBlockEntries;
    BYZ_TRY(ctx->gram_partials.ensure((slab_floats + static_cast<size_t>(grid) * kMaxRows) * sizeof(float)));
    BYZ_TRY(ctx->gram.ensure(static_cast<size_t>(kMaxBlocks) * kBlockEntries * sizeof(double)));
    BYZ_TRY(ctx->gram_rep.ensure(static_cast<size_t>(kMaxRows) * sizeof(int32_t)));
    const int pair_capacity = kMaxRows * (kMaxRows - 1) / 2;
    BYZ_TRY(ctx->near_pairs.ensure(static_cast<size_t>(pair_capacity) * sizeof(int2)));
    const int64_t item_capacity = static_cast<int64_t>(1) << 20;
    BYZ_TRY(ctx->near_partial.ensure(static_cast<size_t>(item_capacity) * sizeof(double)));
    if (ctx->small_sync.ptr == nullptr) {
        BYZ_TRY(ctx->small_sync.ensure(64));
        BYZ_HIP(hipMemsetAsync(ctx->small_sync.ptr, 0, 64, stream));
    }
    float* slabs = ctx->gram_partials.as<float>();
    float* diag_slabs = slabs + slab_floats;
    int32_t* flags = ctx->small_sync.as<int32_t>() + 4;
    if (!ctx->small_configured) {   // per context: the attribute belongs to the (function, device) pair
#define BYZ_ATTR(T, P)                                                                            \
    BYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_gram_kernel<T, P>),          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, kGramLds))
        BYZ_ATTR(true, 0);
        BYZ_ATTR(false, 0);
        BYZ_ATTR(false, 1);
        BYZ_ATTR(false, 2);
        BYZ_ATTR(false, 3);
        BYZ_ATTR(false, 4);
        BYZ_ATTR(false, 5);
        BYZ_ATTR(false, 6);
        BYZ_ATTR(false, 7);
        BYZ_ATTR(false, 8);
#undef BYZ_ATTR
        BYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&small_distance_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kK3Lds));
        ctx->small_configured = true;
    }
    {
        KernelTimer t(ctx, BYZ_K_GRAM, stream);
        const int unrolled = env_int("BYZ_KRUM_SMALL_UNROLL", 1) != 0 && per <= 8 ? static_cast<int>(per) : 0;
#define BYZ_K1(T, P)                                                                               \
    small_gram_kernel<T, P><<<static_cast<unsigned>(T ? 1 : grid), kThreads, kGramLds, stream>>>(    \
        G, n, n_cols, ld, static_cast<int>(T ? 1 : n_slices), slabs, diag_slabs, flags)
        if (n_cols < kSlice) BYZ_K1(true, 0);
        else switch (unrolled) {
            case 1: BYZ_K1(false, 1); break;
            case 2: BYZ_K1(false, 2); break;
            case 3: BYZ_K1(false, 3); break;
            case 4: BYZ_K1(false, 4); break;
            case 5: BYZ_K1(false, 5); break;
            case 6: BYZ_K1(false, 6); break;
            case 7: BYZ_K1(false, 7); break;
            case 8: BYZ_K1(false, 8); break;
            default: BYZ_K1(false, 0); break;
        }
#undef BYZ_K1
        BYZ_TRY(check_launch("small_gram_kernel"));
    }
    {
        KernelTimer t(ctx, BYZ_K_GRAM_REDUCE, stream);
        small_reduce_kernel<<<static_cast<unsigned>(n_blocks * kBlockEntries / 64), kThreads, 0, stream>>>(
            slabs, diag_slabs, grid, n_blocks, n, ctx->gram.as<double>(), dist, flags);
        BYZ_TRY(check_launch("small_reduce_kernel"));
    }
    {
        KernelTimer t(ctx, BYZ_K_DISTANCES, stream);
        DistanceArgs p;
        p.gram = ctx->gram.as<double>();
        p.G = G;
        p.n_rows = n;
        p.n_cols = n_cols;
        p.ld = ld;
        p.dist = dist;
        p.rep = ctx->gram_rep.as<int32_t>();
        p.pairs = ctx->near_pairs.as<int2>();
        p.pair_capacity = pair_capacity;
        p.pair_partial = ctx->near_partial.as<double>();
        p.item_capacity = item_capacity;
        p.sync = ctx->small_sync.as<int32_t>();
        ctx->small_epoch = ctx->small_epoch == 0x7fffffff ? 1 : ctx->small_epoch + 1;
        p.epoch = ctx->small_epoch;
        p.status = device_status_word(ctx);
        // helpers only where a near-duplicate pass could be long enough to need them; all of them must be resident with the
        // worker (one workgroup per CU: 148 KiB of LDS)
        int helpers = env_int("BYZ_KRUM_SMALL_HELPERS", n_cols >= 4096 ? ctx->num_cus / 2 : 0);
        if (helpers > ctx->num_cus - 1) helpers = ctx->num_cus - 1;
        if (helpers < 0) helpers = 0;
        small_distance_kernel<<<static_cast<unsigned>(1 + helpers), kThreads, kK3Lds, stream>>>(p);
        BYZ_TRY(check_launch("small_distance_kernel"));
    }
    return BYZ_OK;
}

int launch_small_distances(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* dist,
                           hipStream_t stream) {
    return small_distances(ctx, G, n_rows, n_cols, ld, dist, stream);
}

// scores (ctx->scores) and the winner (winner_dev) from a distance matrix of n <= 128 rows; out_row (optional): the copy of
// the winning row: K4, K5
int launch_small_select(byz_ctx* ctx, const float* dist, int64_t n_rows, int64_t prefix_len, const float* G, int64_t n_cols,
                        int64_t ld, int32_t* winner_dev, float* out_row, hipStream_t stream) {
    BYZ_REQUIRE(dist && winner_dev && n_rows >= 1 && n_rows <= kMaxRows, "small select: bad arguments");
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(kMaxRows) * sizeof(float)));
    const int n = static_cast<int>(n_rows);
    {
        KernelTimer t(ctx, BYZ_K_ROW_SORT, stream);
        small_score_kernel<<<static_cast<unsigned>((n + 3) / 4), 256, 0, stream>>>(dist, n, static_cast<int>(prefix_len),
                                                                                   ctx->scores.as<float>());
        BYZ_TRY(check_launch("small_score_kernel"));
    }
    {
        KernelTimer t(ctx, BYZ_K_KRUM_ARGMIN, stream);
        const bool copy = out_row != nullptr && G != nullptr;
        int grid = 1;
        if (copy) {
            grid = static_cast<int>(ceil_div(n_cols, 1024));
            if (grid > ctx->num_cus) grid = ctx->num_cus;
            if (grid < 1) grid = 1;
        }
        small_pick_kernel<<<static_cast<unsigned>(grid), 256, 0, stream>>>(ctx->scores.as<float>(), n, G, n_cols, ld, winner_dev,
                                                                           copy ? out_row : nullptr);
        BYZ_TRY(check_launch("small_pick_kernel"));
    }
    return BYZ_OK;
}

}  // namespace byz
