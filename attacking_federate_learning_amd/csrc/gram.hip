// Pairwise client distances, reference defences.py:16-21 (_krum_create_distances).
//
// The reference evaluates ||g_i - g_j|| pair by pair; here the N x N matrix comes from the Gram identity
//     d_ij = sqrt(max(0, c_ii + c_jj - 2 c_ij)),   C = G * G^T,
// because C is a genuine dense contraction over the D parameters and that is what the matrix cores are for.
//
// Arithmetic (two modes, same tiling):
//   exact  v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate): bit-for-bit a k-ordered fmaf chain.  157.3 TF peak.
//   split  every fp32 operand is split EXACTLY into three bf16 planes (the three 8-bit fields of its 24-bit
//          significand) and six v_mfma_f32_32x32x16_bf16 per block stand in for the fp32 product; the three dropped
//          cross terms are below 2^-23 |x y| (measured bias on a sum of squares: -5e-8 relative).  Default for N > 256.
//
// Data layout and tiling (gfx950):
//   * G is row-major (N x D): both operands of C = G G^T are K-contiguous, so every global load is a
//     128-byte row segment (BK = 32 floats).
//   * One workgroup = 4 waves = one 128 x 128 tile of C (lower triangle only, tj <= ti); each wave owns a
//     64 x 64 sub-tile = 2 x 2 MFMA 32 x 32 blocks = 64 accumulator registers.  Two workgroups per CU.
//   * Operand tiles reach LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass),
//     double-buffered, one barrier per stage.  The DMA image is lane-linear, so rows are exactly 32 floats and the
//     bank conflicts of the fragment reads are removed by an XOR swizzle applied to the DMA's SOURCE address and to
//     every read.  Rows that are not 16-byte aligned take the register-staged path (padded 36-float rows).
//   * K schedule.  Few tiles: split-K, one slab per (tile, split), reduced in fp64 in a fixed order.  Many tiles
//     and a long K: chunks of 8192 columns; all chunks of a tile add, in chunk order (a ticket per tile), into
//     the tile's one fp64 slab, and each XCD runs its workgroups in rounds that start together, so the workgroups
//     that share operand row blocks stay in step and hit each other's lines in the XCD's L2.
//
// Numerics:
//   * three accumulation levels: an MFMA chain of at most 2048 columns (256 in split mode, because the bf16
//     MFMA adds into its accumulator with truncation), fp32 sums of a few such chains, fp64 slabs;
//   * c_ii, c_jj and c_ij all come out of the same code with the same k order, so bitwise-identical rows
//     (every malicious client submits the same vector, malicious.py:26-27) give d_ij == 0 exactly; the rows of
//     such a group are then given bitwise identical distance rows (duplicate_rep_kernel), which is what lets the
//     exact ties the reference resolves by visit order survive in either arithmetic.
//
// Algorithmic work per call: N^2 * D flops (half Gram, 2 flops per MAC), 4 N D bytes read.  Bound: MFMA once
// N/4 flop/B exceeds the machine balance (N >~ 80), HBM below that.
#include "common.hpp"

#include <cstdlib>

namespace byz {
namespace {

constexpr int TM = 128;              // tile edge (rows of G per operand tile)
constexpr int BK = 32;               // floats of K per stage
constexpr int LDS_STRIDE = BK + 4;   // 36 floats = 144 B: 16-byte aligned, conflict-free b128 reads
constexpr int THREADS = 256;
constexpr int kFlushK = 2048;        // longest fp32 accumulation chain
constexpr int kLevel1 = 8;           // level-0 chains per level-1 fp32 sum
constexpr int TILE_FLOATS = TM * LDS_STRIDE;
constexpr int kStatusLostTicket = 1;     // bits of the context's sticky device status word
constexpr int kStatusPairOverflow = 2;
constexpr int kStatusFalseTwin = 4;

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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Exact three-way split of eight fp32 values into bf16 planes: x = h + m + l with h, m, l the three 8-bit fields
// of the 24-bit significand (truncation, so every residual is exact and nothing is rounded away).  Products of two
// planes are exact in fp32, which is what lets bf16 MFMAs (16x the fp32 MFMA rate) carry an fp32 contraction.
__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, bf16x8& h, bf16x8& m, bf16x8& l) {
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
    u32x4 hp, mp, lp;
#pragma unroll
    for (int e = 0; e < 4; ++e) {   // pack the high halves of two consecutive values: v_perm_b32
        hp[e] = __builtin_amdgcn_perm(xb[2 * e + 1], xb[2 * e], 0x07060302u);
        mp[e] = __builtin_amdgcn_perm(r1b[2 * e + 1], r1b[2 * e], 0x07060302u);
        lp[e] = __builtin_amdgcn_perm(r2b[2 * e + 1], r2b[2 * e], 0x07060302u);
    }
    h = __builtin_bit_cast(bf16x8, hp);
    m = __builtin_bit_cast(bf16x8, mp);
    l = __builtin_bit_cast(bf16x8, lp);
}

// float offset of (row, 16-byte chunk 0..7) inside one operand tile.
//   register-staged layout: rows padded to 36 floats (conflict-free ds_read_b128);
//   LDS-DMA layout: rows of exactly 32 floats, because `global_load_lds` writes wave-uniform base + 16 * lane and
//   cannot pad; the bank conflicts are removed by an XOR swizzle of the chunk index instead, applied to the
//   SOURCE address of the DMA and to every read (the same involution on both sides).
template <bool DMA>
__device__ __forceinline__ int tile_off(int row, int chunk) {
    return DMA ? row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2) : row * LDS_STRIDE + (chunk << 2);
}

template <typename PartialT, bool DMA, bool SPLIT>
__global__ __launch_bounds__(THREADS, 2) void gram_tile_kernel(const float* __restrict__ G, int64_t n_rows,
                                                               int64_t n_cols, int64_t ld,
                                                               int64_t stages_per_split,
                                                               PartialT* __restrict__ partial, int n_tiles, int slab_tiles,
                                                               const int2* __restrict__ tile_order, int per_xcd, int n_splits,
                                                               int* __restrict__ tickets, int round_size,
                                                               const int32_t* __restrict__ row_index,
                                                               int32_t* __restrict__ device_status) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * TILE_FLOATS];  // [2 buffers][A | B][TM][row]

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
        if (tickets != nullptr) {
            // Chunked schedule: the XCD's workgroups run in rounds of `round_size` (its resident capacity).  A
            // workgroup starts only when every workgroup of the earlier rounds of its XCD has finished, so a round
            // starts together and its members walk K in step: that is what makes them hit each other's operand
            // lines in the XCD's L2 (staggered starts left the hit rate at 28%).  Only scheduling depends on this:
            // earlier workgroups never wait for later ones, and the wait is bounded.
            int* done = tickets + slab_tiles + xcd;
            const int round = round_size > 0 ? seq / round_size : 0;   // 0 disables the gate
            if (round > 0 && threadIdx.x == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * round_size) {
                    __builtin_amdgcn_s_sleep(16);
                    if (++spins > (1u << 20)) break;
                }
            }
            __syncthreads();
            if (split >= n_splits) {
                if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        } else if (split >= n_splits) {
            return;
        }
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int64_t k_begin = static_cast<int64_t>(split) * stages_per_split * BK;
    int64_t k_end = k_begin + stages_per_split * BK;
    if (k_end > n_cols) k_end = n_cols;
    const int n_stages = k_begin < k_end ? static_cast<int>((k_end - k_begin + BK - 1) / BK) : 0;
    const int n_full = k_begin < k_end ? static_cast<int>((k_end - k_begin) / BK) : 0;   // stages without a ragged K tail

    // Rows past the matrix are clamped to the last row: the duplicates land in Gram entries nobody reads, and
    // the main loop carries no per-load branch (a branch around a load makes hipcc drain vmcnt).
    auto row_ptr = [&](int tile_row0, int local_row) {
        int64_t r = static_cast<int64_t>(tile_row0) * TM + local_row;
        if (r > n_rows - 1) r = n_rows - 1;
        if (row_index != nullptr) r = row_index[r];   // logical row r of the (deduplicated) matrix is G[row_index[r]]
        return G + r * ld;
    };
    // register staging (the whole loop without DMA, the ragged K tail with it): 8 lanes cover one 128-byte
    // row segment, 32 rows per pass, 4 passes per operand
    const int ld_chunk = tid & 7;
    const int ld_row = tid >> 3;
    // LDS-DMA staging: wave w moves rows 32w .. 32w+31 of each operand, 8 rows (1 KiB) per instruction;
    // lane l lands at LDS position (row l >> 3, chunk slot l & 7), which holds chunk (l & 7) ^ swizzle(row)
    const float* src_a[4];
    const float* src_b[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if constexpr (DMA) {
            const int r = 32 * wave + 8 * p + (lane >> 3);
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            src_a[p] = row_ptr(ti, r) + 4 * chunk;
            src_b[p] = row_ptr(tj, r) + 4 * chunk;
        } else {
            src_a[p] = row_ptr(ti, ld_row + 32 * p) + 4 * ld_chunk;
            src_b[p] = row_ptr(tj, ld_row + 32 * p) + 4 * ld_chunk;
        }
    }

    f32x4 ra[4], rb[4];
    auto fetch_full = [&](int stage) __attribute__((always_inline)) {   // register path, full stage
        const int64_t k = k_begin + static_cast<int64_t>(stage) * BK;
#pragma unroll
        for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const f32x4u*>(src_a[p] + k);
        if (!diagonal) {
#pragma unroll
            for (int p = 0; p < 4; ++p) rb[p] = *reinterpret_cast<const f32x4u*>(src_b[p] + k);
        }
    };
    auto fetch_tail = [&](int stage) __attribute__((always_inline)) {   // the one ragged stage: zero-fill past k_end
        const int64_t k = k_begin + static_cast<int64_t>(stage) * BK + 4 * ld_chunk;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            ra[p] = load_tail(row_ptr(ti, ld_row + 32 * p) + k, k, k_end);
            if (!diagonal) rb[p] = load_tail(row_ptr(tj, ld_row + 32 * p) + k, k, k_end);
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
        float* A = lds + buf * 2 * TILE_FLOATS;
        float* B = A + TILE_FLOATS;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(A + tile_off<DMA>(ld_row + 32 * p, ld_chunk)) = ra[p];
            if (!diagonal) *reinterpret_cast<f32x4*>(B + tile_off<DMA>(ld_row + 32 * p, ld_chunk)) = rb[p];
        }
    };
    auto dma = [&](int stage, int buf) __attribute__((always_inline)) {
        if constexpr (DMA) {
            const int64_t k = k_begin + static_cast<int64_t>(stage) * BK;
            float* A = lds + buf * 2 * TILE_FLOATS + (32 * wave) * 32;
            float* B = A + TILE_FLOATS;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_a[p] + k),
                                                 (__attribute__((address_space(3))) void*)(A + p * 8 * 32), 16, 0, 0);
            if (!diagonal) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_b[p] + k),
                                                     (__attribute__((address_space(3))) void*)(B + p * 8 * 32), 16, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];
    constexpr bool kWide = sizeof(PartialT) == 8;
    // Long K ranges (kWide): three accumulation levels, so that no fp32 chain is long and no fp64 lives in
    // registers (128 VGPRs of fp64 sums used to push the kernel to 256 VGPRs with spills):
    //   level 0  `acc`   MFMA chain of at most kFlushK = 2048 products,
    //   level 1  `acc2`  fp32 sum of at most kLevel1 = 8 level-0 chains (8 similar-sized terms: ~1e-7),
    //   level 2  the workgroup's own fp64 slab in global memory, read-modify-written every 16,384 columns.
    f32x16 acc2[kWide ? 2 : 1][kWide ? 2 : 1];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.0f;
    if constexpr (kWide) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[m][n][e] = 0.0f;
    }
    // Two schedules share this kernel.
    //   split-K (tickets == nullptr): `split` indexes one of n_splits slabs per tile; gram_reduce_kernel adds them.
    //   chunked (tickets != nullptr): `split` is a K chunk of 8,192 columns and all chunks of a tile accumulate, in
    //   chunk order, into the tile's ONE fp64 slab (ticket per tile, see the epilogue).  Short workgroups that start
    //   together stay within a few stages of each other, so the ~64 workgroups an XCD runs at a time really do
    //   share their 8 + 8 operand row blocks in its L2: with split-K over 322,000 columns (N = 4000, D = 1e7) they
    //   drifted apart and the L2 hit rate was 5% (5.0 TB of fabric reads for a 160 GB matrix).
    const bool chunked = tickets != nullptr;
    PartialT* out = partial + (chunked ? static_cast<int64_t>(tile) : static_cast<int64_t>(split) * slab_tiles + tile) * (TM * TM);
    int level1 = 0;          // level-0 chains summed into acc2 since the last slab update
    bool slab_live = false;  // the slab already holds a partial sum

    const int frag_row = lane & 31;
    const int frag_half = lane >> 5;
    // The bf16 MFMAs add into their fp32 accumulator with truncation, not round-to-nearest (measured: a 2048-column
    // chain of squares comes out 2e-6 low, the fp32-input MFMA 1e-7 either way; the dropped split terms explain only
    // 5e-8).  The bias grows with the chain length, so split mode keeps level-0 chains at 256 columns and lets the
    // round-to-nearest VALU additions of level 1 carry up to 64 of them.
    constexpr int kFlushStages = SPLIT ? 8 : kFlushK / BK;
    constexpr int kLevel1Count = SPLIT ? 64 : kLevel1;   // a chunk of the chunked schedule (8192 columns) never reaches it

    auto compute = [&](int s) __attribute__((always_inline)) {
        const float* A = lds + (s & 1) * 2 * TILE_FLOATS;
        const float* B = diagonal ? A : A + TILE_FLOATS;
        if constexpr (SPLIT) {
            // bf16 x 3: per 16-column step the lane's eight fp32 values of each operand row are split into three
            // exact bf16 planes in registers, and six bf16 MFMAs per 32 x 32 block form
            //   h h' + h m' + m h' + m m' + h l' + l h'      (dropped: m l' + l m' + l l' <= 2^-23 |x y|),
            // smallest terms first.  6 x 32 cycles against 8 x 64 for the fp32 MFMAs of the same 16 columns.
#pragma unroll
            for (int j = 0; j < BK / 16; ++j) {
                bf16x8 ap[3][2], bp[3][2];   // [plane h, m, l][block]
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int row = wr * 64 + m * 32 + frag_row;
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(A + tile_off<DMA>(row, 4 * j + 2 * frag_half));
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(A + tile_off<DMA>(row, 4 * j + 2 * frag_half + 1));
                    split8(lo4, hi4, ap[0][m], ap[1][m], ap[2][m]);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int row = wc * 64 + n * 32 + frag_row;
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(B + tile_off<DMA>(row, 4 * j + 2 * frag_half));
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(B + tile_off<DMA>(row, 4 * j + 2 * frag_half + 1));
                    split8(lo4, hi4, bp[0][n], bp[1][n], bp[2][n]);
                }
                // term-major order: consecutive MFMAs go to the four different accumulators, so none of them
                // waits for the result of the one before it
                constexpr int pa[6] = {2, 0, 1, 1, 0, 0};   // plane of the A operand: l h m m h h
                constexpr int pb[6] = {0, 2, 1, 0, 1, 0};   // plane of the B operand: h l m h m h
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[pa[t]][m], bp[pb[t]][n], acc[m][n], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 a[2], b[2];
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    a[m] = *reinterpret_cast<const f32x4*>(A + tile_off<DMA>(wr * 64 + m * 32 + frag_row, 2 * kk + frag_half));
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    b[n] = *reinterpret_cast<const f32x4*>(B + tile_off<DMA>(wc * 64 + n * 32 + frag_row, 2 * kk + frag_half));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
            }
        }
    };
    auto to_slab = [&](bool last) __attribute__((always_inline)) {
        if constexpr (kWide) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int i = wr * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                        const int j = wc * 64 + n * 32 + (lane & 31);
                        double v = static_cast<double>(acc2[m][n][e]);
                        if (last) v += static_cast<double>(acc[m][n][e]);
                        if (slab_live) v += out[i * TM + j];
                        out[i * TM + j] = v;
                        acc2[m][n][e] = 0.0f;
                    }
            slab_live = true;
            level1 = 0;
        }
    };
    auto flush = [&](int s) __attribute__((always_inline)) {
        if constexpr (kWide) if ((s + 1) % kFlushStages == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc2[m][n][e] += acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
            if (++level1 == kLevel1Count) to_slab(false);
        }
    };

    // prologue: stage 0 into buffer 0
    if (n_stages > 0) {
        if (DMA && n_full > 0) {
            dma(0, 0);
        } else {
            if (n_full > 0) fetch_full(0); else fetch_tail(0);
            stash(0);
        }
    }
    __syncthreads();
    int s = 0;
    for (; s + 1 < n_full; ++s) {   // steady state: the next stage is a full one
        if constexpr (DMA) {
            dma(s + 1, (s + 1) & 1);      // asynchronous: lands in the other buffer while this one multiplies
            compute(s);
        } else {
            fetch_full(s + 1);
            compute(s);
            stash((s + 1) & 1);
        }
        __syncthreads();                  // with a DMA in flight hipcc waits vmcnt(0) here
        flush(s);
    }
    for (; s < n_stages; ++s) {     // at most two stages: the last full one and the ragged tail
        if (s + 1 < n_stages) fetch_tail(s + 1);
        compute(s);
        if (s + 1 < n_stages) stash((s + 1) & 1);
        __syncthreads();
        flush(s);
    }

    if constexpr (kWide) {
        if (chunked) {
            // wait for the previous chunk of this tile (dispatched earlier, so it is running or done), then
            // read-modify-write the slab, then pass the ticket on.  Release/acquire at agent scope: the slab lines
            // may sit in another CU's L1 or be dirty in L2 (MI355X_MICROARCH.md, inter-workgroup visibility).
            bool lost = false;
            if (split > 0) {
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(tickets + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != split) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > (1u << 26)) {   // bounded: a lost ticket must not hang the device ...
                            lost = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                // ... and must not be papered over either: without the ticket the slab is not ours to update.  The
                // launch is marked failed (the host turns the word into BYZ_E_HIP) and this chunk is dropped.
                lost = __syncthreads_or(lost ? 1 : 0) != 0;
                slab_live = true;
            }
            if (lost) {
                if (tid == 0) atomicOr(device_status, kStatusLostTicket);
            } else {
                to_slab(true);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(tickets + tile, split + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (per_xcd > 0)
                    __hip_atomic_fetch_add(tickets + slab_tiles + (blockIdx.x & 7), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            to_slab(true);
        }
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = wr * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const int j = wc * 64 + n * 32 + (lane & 31);
                    out[i * TM + j] = static_cast<PartialT>(acc[m][n][e]);
                }
    }
}

// Q = 1: a block covers 64 columns x 4 rows, one thread per entry.
// Q = 4: a block covers 64 columns of one row, four threads per entry each summing every fourth slab with four
//        loads in flight (many slabs, few tiles: one dependent chain per entry was 50 us at 155 slabs).
// The order of the fp64 additions is fixed either way: the result is deterministic.
template <typename PartialT, int Q>
__global__ __launch_bounds__(256) void gram_reduce_kernel(const PartialT* __restrict__ partial, int n_tiles,
                                                          int splits, int64_t n, double* __restrict__ gram,
                                                          const uint8_t* __restrict__ tile_owned, int accumulate) {
    __shared__ double part[Q == 1 ? 1 : 256];
    const int jl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + jl;
    const int64_t i = Q == 1 ? static_cast<int64_t>(blockIdx.y) * 4 + q : static_cast<int64_t>(blockIdx.y);
    // Q = 4 (few tiles): only the lower triangle is summed -- there the slab reads of consecutive j are consecutive
    // addresses (the upper triangle reads the same slab entries 512 bytes apart) -- and the result is written twice.
    const bool live = i < n && j < n && (Q == 1 || j <= i);
    double s = 0.0;
    bool mine = true;
    if (live) {
        const int64_t hi = i > j ? i : j, lo = i > j ? j : i;
        const int ti = static_cast<int>(hi / TM), tj = static_cast<int>(lo / TM);
        const int tile = ti * (ti + 1) / 2 + tj;
        const int64_t off = static_cast<int64_t>(tile) * (TM * TM) + (hi % TM) * TM + (lo % TM);
        const int64_t slab = static_cast<int64_t>(n_tiles) * (TM * TM);
        // accumulate: the caller's matrix already holds the sum of earlier panels (byz_gram_share_add_dev): this panel's value
        // is added in place -- the same fp64 addition, in the same panel order, a separate N x N `add_` pass did
        mine = tile_owned == nullptr || tile_owned[tile] != 0;
        if (!mine) {
            // a tile of another rank's share: this rank contributes zero to the all-reduced Gram
        } else if (Q == 1) {
            for (int sp = 0; sp < splits; ++sp) s += static_cast<double>(partial[sp * slab + off]);
        } else {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int sp = q;
            for (; sp + 3 * Q < splits; sp += 4 * Q) {
                s0 += static_cast<double>(partial[(sp + 0 * Q) * slab + off]);
                s1 += static_cast<double>(partial[(sp + 1 * Q) * slab + off]);
                s2 += static_cast<double>(partial[(sp + 2 * Q) * slab + off]);
                s3 += static_cast<double>(partial[(sp + 3 * Q) * slab + off]);
            }
            for (; sp < splits; sp += Q) s0 += static_cast<double>(partial[sp * slab + off]);
            s = (s0 + s1) + (s2 + s3);
        }
    }
    if (Q == 1) {
        if (live && !(accumulate && !mine)) gram[i * n + j] = accumulate ? gram[i * n + j] + s : s;
    } else {
        part[threadIdx.x] = s;
        __syncthreads();
        if (q == 0 && live && !(accumulate && !mine)) {
            const double total = (part[jl] + part[64 + jl]) + (part[128 + jl] + part[192 + jl]);
            gram[i * n + j] = accumulate ? gram[i * n + j] + total : total;
            if (j != i) gram[j * n + i] = accumulate ? gram[j * n + i] + total : total;
        }
    }
}

// d_ij from the Gram, and the list of pairs the Gram identity cannot resolve.
//
// c_ii + c_jj - 2 c_ij loses relative accuracy as the rows approach each other: the Gram entries carry ~1e-7 of
// (c_ii + c_jj), so d^2 < eps (c_ii + c_jj) leaves d with a relative error of ~1e-7 / (2 eps).  The reference never has
// that problem, it takes the norm of the difference (defences.py:20).  So every pair i > j below the threshold
// (eps = 1/16: cosine similarity above 0.94) is listed and re-evaluated on the difference itself
// (near_pair_partial_kernel).  After that step a distance is 0 only for rows whose fp32 difference is 0 in every column.
//
// Identical rows would flood that list (the attack makes f of the N clients one vector: f^2 / 2 pairs), so they are
// folded first.  Bitwise identical rows have bitwise identical Gram entries c_ii == c_jj == c_ij (same operands, same
// order, in either arithmetic and through an all-reduce), so rep[i] = the first j < i with that property nominates i's
// representative; ONE pair (i, rep[i]) per folded row goes on the list as the proof, only representatives are paired
// with each other, and the folded rows take their representative's distances afterwards (canonicalise_duplicates).
// A nomination the proof refutes (three doubles equal by coincidence) is reported through the status word.
constexpr double kNearEps = 1.0 / 16.0;

__global__ __launch_bounds__(256) void gram_rep_kernel(const double* __restrict__ gram, int64_t n,
                                                       int32_t* __restrict__ rep) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);   // one wave per row
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const unsigned long long* bits = reinterpret_cast<const unsigned long long*>(gram);
    const unsigned long long cii = bits[i * n + i];
    int best = static_cast<int>(i);
    for (int64_t j0 = 0; j0 < i; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool hit = j < i && bits[i * n + j] == cii && bits[j * n + j] == cii;
        const unsigned long long m = __ballot(hit);
        if (m) {
            best = static_cast<int>(j0) + __builtin_ctzll(m);
            break;
        }
    }
    if (lane == 0) rep[i] = best;
}

// The pair list must have ONE order on every GPU: the multi-GPU paths all-reduce the per-rank sums of squared differences
// element by element (sharded.py), so slot p has to mean the same (i, j) on every rank.  A list filled through
// atomicAdd(pair_count, 1) is ordered by wave timing.  So the list is built in three steps, and its order
// -- ascending i, then ascending j -- is a function of the Gram alone (which the all-reduce leaves bitwise identical on
// every rank): distance_kernel counts the listed pairs of every row (the count does not depend on the order of the atomics),
// pair_offsets_kernel turns the counts into offsets (one workgroup, rows in order), pair_fill_kernel re-evaluates the rows
// that have any and writes their pairs in column order.
// `proven` (optional): dedup.hip's map of this very matrix -- proven[i] == proven[j] <=> rows i and j were compared byte for
// byte and are identical; such a pair needs no second proof on the difference (under the attack that is one pair per
// malicious client: 2399 x 2 rows of D columns read again, 6 ms of the c5s round).
__device__ __forceinline__ bool pair_is_listed(const double* __restrict__ gram, int64_t n, const int32_t* __restrict__ rep,
                                               const int32_t* __restrict__ proven, int64_t i, int64_t j, double cii) {
    const double cjj = gram[j * n + j];
    const double d2 = cii + cjj - 2.0 * gram[i * n + j];
    if (!(d2 < kNearEps * (cii + cjj))) return false;
    if (proven != nullptr && proven[i] == proven[j]) return false;
    const int ri = rep[i], rj = rep[j];
    // representatives pair with each other; a folded row only with its representative (the proof of identity)
    return (ri == i && rj == j) || ri == j;
}

__global__ __launch_bounds__(256) void distance_kernel(const double* __restrict__ gram, int64_t n,
                                                       float* __restrict__ dist, const int32_t* __restrict__ rep,
                                                       const int32_t* __restrict__ proven, int32_t* __restrict__ row_pairs) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
    const int64_t i = static_cast<int64_t>(blockIdx.y) * 4 + (threadIdx.x >> 6);
    if (i >= n || j >= n) return;
    float d;
    if (i == j) {
        d = __builtin_inff();  // the reference keeps no self-distance (defences.py:18-20)
    } else {
        const double cii = gram[i * n + i], cjj = gram[j * n + j];
        const double d2 = cii + cjj - 2.0 * gram[i * n + j];
        // rounding can leave a tiny negative value for near-identical rows; NaN (poisoned input) must stay NaN
        d = static_cast<float>(sqrt(d2 < 0.0 ? 0.0 : d2));
        // rows dedup.hip compared byte for byte are not on the pair list (pair_is_listed), so nothing would correct a d2 that
        // is not EXACTLY zero -- and the Gram gives exact zeros only while both rows went through the same arithmetic with the
        // same scale (the sampled operand split may redo one twin's row block at another shift: ADVICE r3).  Their distance
        // is zero by proof, not by cancellation.
        if (proven != nullptr && proven[i] == proven[j]) d = 0.0f;
        if (i > j && pair_is_listed(gram, n, rep, proven, i, j, cii)) atomicAdd(&row_pairs[i], 1);
    }
    dist[i * n + j] = d;
}

// row_pairs[i] (counts) -> exclusive offsets, in place; *pair_count = the total.  One workgroup: n <= a few 10^4.
__global__ __launch_bounds__(1024) void pair_offsets_kernel(int32_t* __restrict__ row_pairs, int64_t n,
                                                            int32_t* __restrict__ pair_count) {
    __shared__ int32_t wave_total[16];
    __shared__ int32_t carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int32_t c = i < n ? row_pairs[i] : 0;
        int32_t scan = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t up = __shfl_up(scan, d, 64);
            if (lane >= d) scan += up;
        }
        if (lane == 63) wave_total[wave] = scan;
        __syncthreads();
        int32_t before = carry;
        for (int w = 0; w < wave; ++w) before += wave_total[w];
        // saturate instead of wrapping: beyond the capacity the call fails anyway (kStatusPairOverflow)
        const int64_t at = static_cast<int64_t>(before) + scan - c;
        if (i < n) row_pairs[i] = at > 0x3fffffff ? 0x3fffffff : static_cast<int32_t>(at);
        __syncthreads();
        if (threadIdx.x == 1023) {
            const int64_t total = static_cast<int64_t>(before) + scan;
            carry = total > 0x3fffffff ? 0x3fffffff : static_cast<int32_t>(total);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *pair_count = carry;
}

// one wave per row; rows without listed pairs (all of them, unless clients nearly coincide) leave after one load
__global__ __launch_bounds__(256) void pair_fill_kernel(const double* __restrict__ gram, int64_t n,
                                                        const int32_t* __restrict__ rep,
                                                        const int32_t* __restrict__ proven,
                                                        const int32_t* __restrict__ row_pairs,
                                                        const int32_t* __restrict__ pair_count, int2* __restrict__ pairs,
                                                        int pair_capacity) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    int at = row_pairs[i];
    const int stop = i + 1 < n ? row_pairs[i + 1] : *pair_count;
    if (stop == at) return;
    const double cii = gram[i * n + i];
    for (int64_t j0 = 0; j0 < i && at < stop; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool listed = j < i && pair_is_listed(gram, n, rep, proven, i, j, cii);
        const unsigned long long m = __ballot(listed);
        if (listed) {
            const int slot = at + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            if (slot < pair_capacity) pairs[slot] = make_int2(static_cast<int>(i), static_cast<int>(j));
        }
        at += __builtin_popcountll(m);
    }
}

// sum over a column chunk of (g_i - g_j)^2: the difference in fp32 as the reference forms it, squares and sums in
// fp64.  Work item w = pair * n_chunks + chunk, persistent grid, one partial per item (fixed order: reproducible).
constexpr int kPairChunk = 32768;
__global__ __launch_bounds__(256) void near_pair_partial_kernel(const float* __restrict__ G, int64_t n_cols, int64_t ld,
                                                                const int32_t* __restrict__ row_index,
                                                                const int2* __restrict__ pairs,
                                                                const int32_t* __restrict__ pair_count, int pair_capacity,
                                                                int n_chunks, int64_t item_capacity,
                                                                double* __restrict__ partial, int32_t* __restrict__ status) {
    __shared__ double red[256];
    int count = *pair_count;
    if (count > pair_capacity || static_cast<int64_t>(count) * n_chunks > item_capacity) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(status, kStatusPairOverflow);
        return;   // the caller gets an error, not a half-patched matrix
    }
    const int64_t items = static_cast<int64_t>(count) * n_chunks;
    const bool vec = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0);
    for (int64_t w = blockIdx.x; w < items; w += gridDim.x) {
        const int2 pr = pairs[w / n_chunks];
        const int64_t k0 = (w % n_chunks) * kPairChunk;
        const int64_t k1 = k0 + kPairChunk < n_cols ? k0 + kPairChunk : n_cols;
        const float* a = G + static_cast<int64_t>(row_index ? row_index[pr.x] : pr.x) * ld;
        const float* b = G + static_cast<int64_t>(row_index ? row_index[pr.y] : pr.y) * ld;
        double acc = 0.0;
        if (vec) {
            const int64_t kv = k0 + ((k1 - k0) & ~static_cast<int64_t>(3));
            for (int64_t k = k0 + 4 * threadIdx.x; k < kv; k += 4 * 256) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(a + k), y = *reinterpret_cast<const f32x4*>(b + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double df = static_cast<double>(__fsub_rn(x[e], y[e]));
                    acc = fma(df, df, acc);
                }
            }
            for (int64_t k = kv + threadIdx.x; k < k1; k += 256) {
                const double df = static_cast<double>(__fsub_rn(a[k], b[k]));
                acc = fma(df, df, acc);
            }
        } else {
            for (int64_t k = k0 + threadIdx.x; k < k1; k += 256) {
                const double df = static_cast<double>(__fsub_rn(a[k], b[k]));
                acc = fma(df, df, acc);
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[w] = red[0];
        __syncthreads();
    }
}

// sq[p] = sum of the pair's chunk partials, in chunk order
__global__ __launch_bounds__(256) void near_pair_sum_kernel(const double* __restrict__ partial,
                                                            const int32_t* __restrict__ pair_count, int pair_capacity,
                                                            int n_chunks, int64_t item_capacity, double* __restrict__ sq) {
    const int count = *pair_count;
    if (count > pair_capacity || static_cast<int64_t>(count) * n_chunks > item_capacity) return;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < count; p += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int c = 0; c < n_chunks; ++c) s += partial[static_cast<int64_t>(p) * n_chunks + c];
        sq[p] = s;
    }
}

__global__ __launch_bounds__(256) void near_pair_apply_kernel(const double* __restrict__ sq, const int2* __restrict__ pairs,
                                                              const int32_t* __restrict__ pair_count, int pair_capacity,
                                                              int64_t n, float* __restrict__ dist,
                                                              const int32_t* __restrict__ rep, int32_t* __restrict__ status) {
    const int count = *pair_count;
    if (count > pair_capacity) return;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < count; p += gridDim.x * blockDim.x) {
        const int2 pr = pairs[p];
        const float d = static_cast<float>(sqrt(sq[p]));
        // a row folded into pr.y whose difference from it is not zero after all: its other pairs were never listed
        if (rep[pr.x] == pr.y && sq[p] != 0.0) atomicOr(status, kStatusFalseTwin);
        dist[static_cast<int64_t>(pr.x) * n + pr.y] = d;
        dist[static_cast<int64_t>(pr.y) * n + pr.x] = d;
    }
}

// Identical rows (every malicious client submits the same vector, malicious.py:26-27) must keep bitwise identical
// distance rows, because the reference resolves their exactly tied Krum scores by visit order.  The fp32-input MFMA
// gives that for free (a * b is computed the same way whichever operand a row is); the bf16 x 3 arithmetic does not
// (its six terms are accumulated in an order that depends on which operand a row is), although it still gives
// d_pq == 0 exactly for identical rows p, q.  So duplicates are recognised by d == 0 and every member of a group
// takes the distances of the group's first row: rep[i] = the smallest j with d_ij == 0 (or i itself).
__global__ __launch_bounds__(256) void duplicate_rep_kernel(const float* __restrict__ dist, int64_t n,
                                                            int32_t* __restrict__ rep, int32_t* __restrict__ any) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);   // one wave per row
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    int best = static_cast<int>(i);
    for (int64_t j0 = 0; j0 < i; j0 += 64) {
        const int64_t j = j0 + lane;
        const bool hit = j < i && dist[i * n + j] == 0.0f;
        const unsigned long long m = __ballot(hit);
        if (m) {
            best = static_cast<int>(j0) + __builtin_ctzll(m);
            break;
        }
    }
    if (lane == 0) {
        rep[i] = best;
        if (best != static_cast<int>(i)) *any = 1;
    }
}

// rep[] chains (i -> j -> k when "zero distance" came from cancellation rather than identity and is not transitive) are
// followed to their root, so that the copy below only ever reads rows and columns nobody writes.
__global__ __launch_bounds__(1024) void duplicate_compress_kernel(int64_t n, int32_t* __restrict__ rep,
                                                                  const int32_t* __restrict__ any) {
    if (*any == 0) return;
    for (int round = 0; round < 16; ++round) {   // rep[i] < i along a chain: pointer jumping converges in log2(n) rounds
        bool changed = false;
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const int32_t r = rep[i], rr = rep[r];
            if (rr != r) {
                rep[i] = rr;
                changed = true;
            }
        }
        if (!__syncthreads_or(changed ? 1 : 0)) break;
    }
}

__global__ __launch_bounds__(256) void duplicate_copy_kernel(float* __restrict__ dist, int64_t n,
                                                             const int32_t* __restrict__ rep,
                                                             const int32_t* __restrict__ any) {
    if (*any == 0) return;   // no duplicates: the matrix stays as it is
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 64 + (threadIdx.x & 63);
    const int64_t i = static_cast<int64_t>(blockIdx.y) * 4 + (threadIdx.x >> 6);
    if (i >= n || j >= n || i == j) return;
    const int ri = rep[i], rj = rep[j];
    if (ri == i && rj == j) return;            // both are representatives: nobody writes this entry
    dist[i * n + j] = ri == rj ? 0.0f : dist[static_cast<int64_t>(ri) * n + rj];
}

int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : fallback;
}

}  // namespace

// Gram of the n_rows logical rows G[row_index[r]] (row_index == nullptr: the rows themselves)
// share_count / share_index: this launch computes only every share_count-th tile of the (XCD-friendly) tile list,
// starting with share_index, and writes zeros for the others: W ranks that hold the same rows each take one share and
// the sum of their outputs is the Gram (the client-sharded multi-GPU path, sharded.py).
static int launch_gram_rows(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                            const int32_t* row_index, double* gram, hipStream_t stream, int share_count = 1,
                            int share_index = 0, bool accumulate = false) {
    BYZ_REQUIRE(G && gram && n_rows > 0 && n_cols > 0 && ld >= n_cols, "gram: bad shape %lld x %lld ld %lld",
                (long long)n_rows, (long long)n_cols, (long long)ld);
    BYZ_REQUIRE(share_count >= 1 && share_index >= 0 && share_index < share_count, "gram: bad share %d of %d",
                share_index, share_count);
    const int64_t T = ceil_div(n_rows, TM);
    const int64_t n_tiles_all = T * (T + 1) / 2;
    if (n_tiles_all > 0x7fffffff) {
        set_error("gram: too many tiles");
        return BYZ_E_UNSUPPORTED;
    }
    // tiles of this share: list positions share_index, share_index + share_count, ...
    const int64_t n_tiles = (n_tiles_all - share_index + share_count - 1) / share_count;
    if (n_tiles <= 0) {   // more ranks than tiles: nothing to compute here
        if (!accumulate) BYZ_HIP(hipMemsetAsync(gram, 0, static_cast<size_t>(n_rows) * n_rows * sizeof(double), stream));
        return BYZ_OK;
    }
    const int64_t stages = ceil_div(n_cols, BK);
    // Everything that decides the ARITHMETIC (mode, split count, chunking) is derived from the tile count of the WHOLE
    // triangle, never from this rank's share of it: (i) shares differ by one tile between ranks, and ranks that summed in
    // different orders would give identical rows' c_ii and c_ij (owned by different ranks) different bits -- which defeats
    // gram_rep's folding of identical rows and floods the near-pair list; (ii) the sum of the W shares then is
    // BITWISE the Gram one GPU computes alone (tests/test_gpu_sharded.py), so sharded and unsharded runs select alike.
    const int64_t plan_tiles = n_tiles_all;
    // split-K.  Two workgroups fit a CU (LDS), so the chip runs `slots` workgroups at a time; the grid is
    // n_tiles * splits of them, all of equal length.  Pick the split count whose last round of workgroups
    // is (nearly) full -- 528 tiles x 2 splits would leave the chip one third idle, 528 x 31 does not --
    // while keeping at least `min_stages` K stages (512 columns) per slab so that slab traffic stays a small
    // fraction of the matrix traffic.
    const int64_t slots = static_cast<int64_t>(ctx->num_cus) * 2;
    // few tiles and a short K (the reference's own sizes: N = 100, D = 79,510 is ONE tile): allow slabs of 128
    // columns so that the tile count x slab count still covers the chip
    int64_t min_stages = env_int("BYZ_GRAM_MIN_STAGES", 0);
    if (min_stages <= 0) min_stages = plan_tiles * (stages / 16) < slots ? 4 : 16;
    int64_t max_splits = stages / min_stages;
    if (max_splits < 1) max_splits = 1;
    if (max_splits > 4096) max_splits = 4096;
    int64_t splits = 1;
    {
        double best = -1.0;
        const int64_t want = ceil_div(slots * 3, plan_tiles);   // at least ~3 rounds when K allows it
        for (int64_t s = 1; s <= max_splits; ++s) {
            const int64_t wgs = plan_tiles * s;
            const double eff = static_cast<double>(wgs) / static_cast<double>(ceil_div(wgs, slots) * slots);
            // prefer fuller last rounds; among equals the fewer slabs; below `want` only if nothing else fits
            const double score = eff - (s < want ? 0.05 : 0.0) - 1e-4 * static_cast<double>(s > want ? s - want : 0);
            if (score > best) {
                best = score;
                splits = s;
            }
        }
    }
    // one tile (N <= 128, the reference's own sizes): one workgroup per CU.  More slabs cost more in slab traffic
    // and in the reduction than they gain in streaming parallelism (measured at N = 100, D = 79,510: 85 us per Krum
    // round with 256 slabs, 100 us with 512, 97 us with 128)
    if (plan_tiles == 1 && splits > ctx->num_cus) splits = ctx->num_cus;
    const int forced = env_int("BYZ_GRAM_SPLITS", 0);
    if (forced > 0) splits = forced;
    if (splits < 1) splits = 1;
    if (splits > stages) splits = stages;
    if (splits > 65535) splits = 65535;
    // Arithmetic of the contraction (BYZ_GRAM_MODE overrides):
    //   exact   fp32-input MFMA, bit-for-bit an fmaf chain; the default while the problem is at most two tiles wide
    //           (N <= 256), where the kernel is launch/HBM bound anyway;
    //   split   bf16 x 3: every fp32 value is split exactly into three bf16 planes and six bf16 MFMAs per block stand
    //           in for the fp32 product.  Default for N > 256.
    const char* mode_env = std::getenv("BYZ_GRAM_MODE");
    const std::string mode_s = mode_env ? mode_env : (n_tiles_all >= 4 ? "split" : "exact");
    const bool dma = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0) && env_int("BYZ_GRAM_NO_DMA", 0) == 0;
    const bool split_mode = dma && (mode_s == "split" || mode_s == "f16x2");   // f16x2 exists only on pre-split operands
    // chunked schedule (see the kernel): many tiles and a long K
    const int64_t chunk_stages = env_int("BYZ_GRAM_CHUNK_COLS", 8192) / BK;
    const bool chunked = plan_tiles >= 256 && stages > 2 * chunk_stages && env_int("BYZ_GRAM_NO_CHUNKS", 0) == 0 &&
                         chunk_stages * BK <= 8192;   // a chunk must end before level 1 spills to the slab
    if (chunked) splits = ceil_div(stages, chunk_stages);
    const int64_t stages_per_split = chunked ? chunk_stages : ceil_div(stages, splits);
    splits = ceil_div(stages, stages_per_split);
    // the fp32 slab format is only for K ranges that fit ONE level-0 chain
    const bool wide = chunked || stages_per_split * BK > (split_mode ? 256 : kFlushK);
    const size_t slab = static_cast<size_t>(TM) * TM * (wide ? sizeof(double) : sizeof(float));
    BYZ_TRY(ctx->gram_partials.ensure(static_cast<size_t>(chunked ? 1 : splits) * n_tiles_all * slab));
    int* tickets = nullptr;
    if (chunked) {
        BYZ_TRY(ctx->gram_tickets.ensure(static_cast<size_t>(n_tiles_all + 8) * sizeof(int)));   // + one finished-count per XCD
        tickets = ctx->gram_tickets.as<int>();
        BYZ_HIP(hipMemsetAsync(tickets, 0, static_cast<size_t>(n_tiles_all + 8) * sizeof(int), stream));
    }
    // tile list in 8 x 8 super-block order (see the kernel); rebuilt only when the tile count or the share changes
    if (ctx->tile_order_T != T || ctx->tile_order_share != share_count * 65536 + share_index) {
        ctx->tile_order_host.clear();
        std::vector<uint8_t> owned(static_cast<size_t>(n_tiles_all), 0);
        const int64_t S = ceil_div(T, 8);
        int64_t position = 0;
        for (int64_t I = 0; I < S; ++I)
            for (int64_t J = 0; J <= I; ++J)
                for (int64_t ti = I * 8; ti < I * 8 + 8 && ti < T; ++ti)
                    for (int64_t tj = J * 8; tj < J * 8 + 8 && tj <= ti; ++tj) {
                        if (position++ % share_count != share_index) continue;
                        ctx->tile_order_host.push_back(static_cast<int32_t>(ti));
                        ctx->tile_order_host.push_back(static_cast<int32_t>(tj));
                        owned[static_cast<size_t>(ti * (ti + 1) / 2 + tj)] = 1;
                    }
        BYZ_TRY(ctx->tile_order.ensure(ctx->tile_order_host.size() * sizeof(int32_t)));
        BYZ_TRY(ctx->tile_owned.ensure(owned.size()));
        BYZ_HIP(hipMemcpyAsync(ctx->tile_order.ptr, ctx->tile_order_host.data(),
                               ctx->tile_order_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BYZ_HIP(hipMemcpyAsync(ctx->tile_owned.ptr, owned.data(), owned.size(), hipMemcpyHostToDevice, stream));
        BYZ_HIP(hipStreamSynchronize(stream));   // the host vectors are pageable; once per matrix height and share
        ctx->tile_order_T = T;
        ctx->tile_order_share = share_count * 65536 + share_index;
    }
    const uint8_t* tile_owned = share_count > 1 ? ctx->tile_owned.as<uint8_t>() : nullptr;
    // XCD-partitioned order only when every XCD gets enough tiles for its shares to be even (<= 3% apart)
    const bool partitioned = n_tiles >= 256;
    const int64_t per_xcd = partitioned ? ceil_div(n_tiles, 8) : 0;
    const int64_t grid_wgs = partitioned ? 8 * per_xcd * splits : n_tiles * splits;
    if (grid_wgs > 0x7fffffff) {
        set_error("gram: grid too large");
        return BYZ_E_UNSUPPORTED;
    }
    // long K and many tiles: the operands are split ONCE into 16-bit planes (gram_planes.hip) instead of once per tile.
    //   BYZ_GRAM_MODE unset / f16x2: two fp16 planes, three MFMAs per block (error 6e-8, see gram_planes.hip);
    //   BYZ_GRAM_MODE=split:         three bf16 planes, bitwise the fused split kernel's slabs
    const bool planes_ok = chunked && dma && chunk_stages * BK == 8192 && gram_planes_enabled();
    const bool f16 = planes_ok && (mode_env == nullptr || mode_s == "f16x2");
    const bool planes = planes_ok && (f16 || split_mode);
    if (planes) {
        std::vector<uint8_t> owned;
        if (share_count > 1) owned.assign(static_cast<size_t>(n_tiles_all), 0);
        BYZ_TRY(launch_gram_planes(ctx, G, n_rows, n_cols, ld, row_index, ctx->gram_partials.as<double>(), share_count,
                                   share_index, share_count > 1 ? owned.data() : nullptr, f16, stream));
        if (share_count > 1) {
            BYZ_TRY(ctx->tile_owned.ensure(owned.size()));
            BYZ_HIP(hipMemcpyAsync(ctx->tile_owned.ptr, owned.data(), owned.size(), hipMemcpyHostToDevice, stream));
            BYZ_HIP(hipStreamSynchronize(stream));   // pageable host vector
            ctx->tile_order_T = -1;                   // tile_owned no longer describes the fused kernel's list
        }
    } else {
        KernelTimer t(ctx, BYZ_K_GRAM, stream);
        const int2* order = ctx->tile_order.as<int2>();
        const int round_size = env_int("BYZ_GRAM_ROUND", ctx->num_cus / 8 * 2);   // workgroups an XCD holds at a time
        const unsigned grid = static_cast<unsigned>(grid_wgs);   // (dma: global_load_lds moves 16 bytes per lane, so every row segment must be 16-byte aligned)
#define BYZ_GRAM(T, D, S)                                                                                     \
    gram_tile_kernel<T, D, S><<<grid, THREADS, 0, stream>>>(G, n_rows, n_cols, ld, stages_per_split,           \
                                                            ctx->gram_partials.as<T>(), (int)n_tiles,           \
                                                            (int)n_tiles_all, order,                             \
                                                            (int)per_xcd, (int)splits, tickets, round_size, row_index,   \
                                                            device_status_word(ctx))
        if (wide) {
            if (split_mode) BYZ_GRAM(double, true, true);
            else if (dma) BYZ_GRAM(double, true, false);
            else BYZ_GRAM(double, false, false);
        } else {
            if (split_mode) BYZ_GRAM(float, true, true);
            else if (dma) BYZ_GRAM(float, true, false);
            else BYZ_GRAM(float, false, false);
        }
#undef BYZ_GRAM
        BYZ_TRY(check_launch("gram_tile_kernel"));
    }
    {
        KernelTimer t(ctx, BYZ_K_GRAM_REDUCE, stream);
        // four threads per entry only when entries alone cannot fill the chip (the 128 x 128 of one tile)
        const bool many = !chunked && splits >= 16 && n_rows * n_rows <= (1 << 18);
        dim3 grid(static_cast<unsigned>(ceil_div(n_rows, 64)), static_cast<unsigned>(many ? n_rows : ceil_div(n_rows, 4)));
#define BYZ_REDUCE(T, Q) gram_reduce_kernel<T, Q><<<grid, 256, 0, stream>>>(ctx->gram_partials.as<T>(), (int)n_tiles_all, (int)(chunked ? 1 : splits), n_rows, gram, tile_owned, accumulate ? 1 : 0)
        if (wide) {
            if (many) BYZ_REDUCE(double, 4); else BYZ_REDUCE(double, 1);
        } else {
            if (many) BYZ_REDUCE(float, 4); else BYZ_REDUCE(float, 1);
        }
#undef BYZ_REDUCE
        BYZ_TRY(check_launch("gram_reduce_kernel"));
    }
    return BYZ_OK;
}

int launch_gram(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, double* gram,
                hipStream_t stream) {
    BYZ_REQUIRE(G && gram && n_rows > 0 && n_cols > 0 && ld >= n_cols, "gram: bad shape %lld x %lld ld %lld",
                (long long)n_rows, (long long)n_cols, (long long)ld);
    ctx->row_map_rows = 0;
    // Identical rows first (dedup.hip): worth a look once the Gram is compute bound and a 4-byte read-back does not
    // show (N >= 512); used when it removes at least one row of tiles.
    if (n_rows >= 512 && env_int("BYZ_GRAM_DEDUP", 1) != 0) {
        int64_t n_unique = n_rows;
        BYZ_TRY(find_unique_rows(ctx, G, n_rows, n_cols, ld, stream, &n_unique));
        ctx->row_map_rows = n_rows;   // row_map[i] == row_map[j]  <=>  rows i and j are bitwise identical
        if (ceil_div(n_unique, TM) < ceil_div(n_rows, TM)) {
            BYZ_TRY(ctx->gram_compact.ensure(static_cast<size_t>(n_unique) * n_unique * sizeof(double)));
            double* compact = ctx->gram_compact.as<double>();
            BYZ_TRY(launch_gram_rows(ctx, G, n_unique, n_cols, ld, ctx->unique_rows.as<int32_t>(), compact, stream));
            BYZ_TRY(launch_gram_expand(ctx, compact, n_unique, n_rows, gram, stream));
            return BYZ_OK;
        }
    }
    return launch_gram_rows(ctx, G, n_rows, n_cols, ld, nullptr, gram, stream);
}

int launch_gram_share(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                      int share_count, int share_index, double* gram, hipStream_t stream, bool accumulate) {
    ctx->row_map_rows = 0;
    return launch_gram_rows(ctx, G, n_rows, n_cols, ld, row_index, gram, stream, share_count, share_index, accumulate);
}

// Identical rows must end up with bitwise identical distance rows, because the reference resolves their exactly tied
// scores by visit order: rep[i] = the smallest j with d_ij == 0, every member of a group takes the group's first row.
static int canonicalise_duplicates(byz_ctx* ctx, int64_t n, float* dist, hipStream_t stream) {
    BYZ_TRY(ctx->dup_rep.ensure(static_cast<size_t>(n + 1) * sizeof(int32_t)));
    int32_t* rep = ctx->dup_rep.as<int32_t>();
    BYZ_HIP(hipMemsetAsync(rep + n, 0, sizeof(int32_t), stream));
    duplicate_rep_kernel<<<static_cast<unsigned>(ceil_div(n, 4)), 256, 0, stream>>>(dist, n, rep, rep + n);
    BYZ_TRY(check_launch("duplicate_rep_kernel"));
    duplicate_compress_kernel<<<1, 1024, 0, stream>>>(n, rep, rep + n);
    BYZ_TRY(check_launch("duplicate_compress_kernel"));
    const dim3 grid(static_cast<unsigned>(ceil_div(n, 64)), static_cast<unsigned>(ceil_div(n, 4)));
    duplicate_copy_kernel<<<grid, 256, 0, stream>>>(dist, n, rep, rep + n);
    return check_launch("duplicate_copy_kernel");
}

static int near_pair_chunks(int64_t n_cols) { return static_cast<int>(ceil_div(n_cols, kPairChunk)); }

static int ensure_pair_buffers(byz_ctx* ctx, int64_t n) {
    // the list is capped: beyond kPairCapacity near-duplicate pairs the call fails loudly (status word) instead of
    // working through what is by then an O(N^2 D) problem -- the reference's own cost
    int64_t cap = n * (n - 1) / 2;
    const int64_t limit = env_int("BYZ_NEAR_PAIR_CAPACITY", 1 << 20);
    if (cap > limit) cap = limit;
    if (cap < 1) cap = 1;
    BYZ_TRY(ctx->near_pairs.ensure(static_cast<size_t>(cap) * sizeof(int2)));
    BYZ_TRY(ctx->near_sq.ensure(static_cast<size_t>(cap) * sizeof(double)));
    ctx->near_pair_capacity = cap;
    return BYZ_OK;
}

// sq_dev[p] = sum over THIS matrix's columns of (g_i - g_j)^2 for every listed pair (the column-sharded path adds the
// ranks' vectors up before applying them)
int launch_near_pair_sqdist(byz_ctx* ctx, const float* G, int64_t n_cols, int64_t ld, const int32_t* row_index,
                            double* sq_dev, hipStream_t stream) {
    const int n_chunks = near_pair_chunks(n_cols);
    const int64_t item_capacity = static_cast<int64_t>(1) << 22;
    BYZ_TRY(ctx->near_partial.ensure(static_cast<size_t>(item_capacity) * sizeof(double)));
    KernelTimer t(ctx, BYZ_K_DISTANCES, stream);
    near_pair_partial_kernel<<<static_cast<unsigned>(ctx->num_cus * 8), 256, 0, stream>>>(
        G, n_cols, ld, row_index, ctx->near_pairs.as<int2>(), near_pair_count_word(ctx), (int)ctx->near_pair_capacity, n_chunks,
        item_capacity, ctx->near_partial.as<double>(), device_status_word(ctx));
    BYZ_TRY(check_launch("near_pair_partial_kernel"));
    near_pair_sum_kernel<<<64, 256, 0, stream>>>(ctx->near_partial.as<double>(), near_pair_count_word(ctx),
                                                 (int)ctx->near_pair_capacity, n_chunks, item_capacity, sq_dev);
    return check_launch("near_pair_sum_kernel");
}

int launch_near_pair_apply(byz_ctx* ctx, const double* sq_dev, int64_t n, float* dist, hipStream_t stream) {
    {
        KernelTimer t(ctx, BYZ_K_DISTANCES, stream);
        near_pair_apply_kernel<<<64, 256, 0, stream>>>(sq_dev, ctx->near_pairs.as<int2>(), near_pair_count_word(ctx),
                                                       (int)ctx->near_pair_capacity, n, dist, ctx->gram_rep.as<int32_t>(),
                                                       device_status_word(ctx));
        BYZ_TRY(check_launch("near_pair_apply_kernel"));
    }
    return canonicalise_duplicates(ctx, n, dist, stream);
}

// Distances from a Gram matrix.  With G (the rows the Gram was taken over) the near-duplicate pairs are re-evaluated
// on the difference itself and the result matches the reference's norm-of-difference to fp32 rounding for every pair;
// without G (the column-sharded path: a rank sees only its slice) the pairs are listed in the context and the caller
// finishes with launch_near_pair_sqdist / an all-reduce / launch_near_pair_apply.
// rows_canonical: the caller guarantees identical rows already have bitwise identical distance rows and exact zeros
// between them (exact-arithmetic Gram of this very call, no near pairs possible to list without G) -- NEVER inferred.
int launch_distances_from_gram(byz_ctx* ctx, const double* gram, int64_t n, float* dist, hipStream_t stream,
                               const float* G, int64_t n_cols, int64_t ld) {
    BYZ_REQUIRE(gram && dist && n > 0, "distances: bad arguments");
    BYZ_TRY(ensure_pair_buffers(ctx, n));
    BYZ_TRY(ctx->gram_rep.ensure(static_cast<size_t>(n) * sizeof(int32_t)));
    BYZ_TRY(ctx->near_rows.ensure(static_cast<size_t>(n) * sizeof(int32_t)));
    int32_t* row_pairs = ctx->near_rows.as<int32_t>();
    BYZ_HIP(hipMemsetAsync(row_pairs, 0, static_cast<size_t>(n) * sizeof(int32_t), stream));
    {
        KernelTimer t(ctx, BYZ_K_DISTANCES, stream);
        gram_rep_kernel<<<static_cast<unsigned>(ceil_div(n, 4)), 256, 0, stream>>>(gram, n, ctx->gram_rep.as<int32_t>());
        BYZ_TRY(check_launch("gram_rep_kernel"));
        dim3 grid(static_cast<unsigned>(ceil_div(n, 64)), static_cast<unsigned>(ceil_div(n, 4)));
        // identical rows already proven by dedup.hip for THIS matrix (launch_gram sets row_map_rows; every other producer of a
        // Gram resets it)
        const int32_t* proven = ctx->row_map_rows == n && ctx->row_map.ptr != nullptr ? ctx->row_map.as<int32_t>() : nullptr;
        distance_kernel<<<grid, 256, 0, stream>>>(gram, n, dist, ctx->gram_rep.as<int32_t>(), proven, row_pairs);
        BYZ_TRY(check_launch("distance_kernel"));
        // the pair list in its canonical order (ascending i, then j): see pair_is_listed
        pair_offsets_kernel<<<1, 1024, 0, stream>>>(row_pairs, n, near_pair_count_word(ctx));
        BYZ_TRY(check_launch("pair_offsets_kernel"));
        pair_fill_kernel<<<static_cast<unsigned>(ceil_div(n, 4)), 256, 0, stream>>>(
            gram, n, ctx->gram_rep.as<int32_t>(), proven, row_pairs, near_pair_count_word(ctx), ctx->near_pairs.as<int2>(),
            (int)ctx->near_pair_capacity);
        BYZ_TRY(check_launch("pair_fill_kernel"));
    }
    if (G != nullptr) {
        BYZ_TRY(launch_near_pair_sqdist(ctx, G, n_cols, ld, nullptr, ctx->near_sq.as<double>(), stream));
        return launch_near_pair_apply(ctx, ctx->near_sq.as<double>(), n, dist, stream);
    }
    // Without G the matrix is NOT canonicalised here: a zero that the Gram identity produced by cancellation (two clients
    // that nearly coincide) is not an identity, and folding such a row into its neighbour would overwrite its whole
    // distance row with the neighbour's before the caller's near-pair step could correct the one entry (found by
    // tests/test_gpu_sharded.py in round 3: distances of a near-duplicate row off by 4e-6).  Every zero -- true or not --
    // is on the pair list (0 < eps (c_ii + c_jj)), so a caller that sees byz_near_pairs_count() == 0 has no identical rows
    // to canonicalise, and any other caller must finish with byz_near_pairs_apply_dev, which canonicalises on proven zeros.
    return BYZ_OK;
}

}  // namespace byz
