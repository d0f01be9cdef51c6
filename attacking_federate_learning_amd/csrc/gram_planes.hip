// The Gram contraction of defences.py:16-21 on PRE-SPLIT operands (the long-K, many-tile case: N >= 2817, D > 16384).
//
// gram.hip's split arithmetic turns every fp32 operand into three exact bf16 planes and feeds six bf16 MFMAs per block.
// There the split runs inside the tile kernel, i.e. once per (tile, element): at N = 4000 every element of G is split
// ~32 times, and that VALU work (and its power) is what holds the MFMA pipe at a third of its rate.  Here the split
// runs ONCE per element:
//
//   plane_split_kernel   G[:, k0 : k0 + SC] (fp32, row-major)  ->  three bf16 planes of that column range, stored in
//                        MFMA FRAGMENT ORDER: for every (32-row block, 16-column step, plane) the 1 KiB image that the
//                        64 lanes of a wave hold as the A/B operand of v_mfma_f32_32x32x16_bf16 (lane l: row l & 31,
//                        columns 8 (l >> 5) .. + 7).  HBM bound: 4 bytes in, 6 bytes out per element.
//   gram_planes_kernel   one workgroup = 8 waves = a 256 x 128 tile of C = G G^T (two 128 x 128 slabs of gram.hip's
//                        slab format), each wave a 64 x 64 sub-tile.  A stage (32 columns) of the 12 row blocks is
//                        72 KiB and reaches LDS by LDS-DMA as 72 contiguous 1 KiB pieces (lane-linear on both sides:
//                        perfectly coalesced, conflict-free ds_read_b128, no swizzle); two buffers = 144 of the 160 KiB.
//                        The inner loop is ds_read + MFMA only.
//
// The super-chunk SC (a multiple of the 8192-column chunk) is sized by a memory budget; per super-chunk one split launch
// and one tile launch, stream-ordered.  Inside a launch the schedule is gram.hip's chunked one: chunks of 8192 columns,
// all chunks of a tile add in chunk order into the tile's fp64 slabs (ticket per tile), XCD-partitioned tile list in
// super-block order, rounds that start together.
//
// Arithmetic is EXACTLY gram.hip's split mode, operation for operation (same six terms in the same order per 16-column
// step, 256-column MFMA chains, fp32 level-1 sums, fp64 slab per chunk), so the Gram is bitwise identical to the one the
// fused kernel produces; tests/test_gpu_scale.py::test_plane_gram_is_bitwise_the_fused_gram holds it to that.
#include "common.hpp"

#include <cstdlib>

namespace byz {
namespace {

constexpr int kWgRows = 256;            // rows of the workgroup tile (A side)
constexpr int kWgCols = 128;            // columns of the workgroup tile (B side) = one slab edge
constexpr int kSlab = 128;              // slab edge of gram.hip (TM)
constexpr int kThreads = 512;
constexpr int kStageCols = 32;          // two 16-column MFMA steps
constexpr int kFragBytes = 1024;        // 64 lanes x 16 bytes
constexpr int kRowBlocks = (kWgRows + kWgCols) / 32;  // 8 A + 4 B
constexpr int kChunkStages = 256;       // 8192 columns: must match gram.hip's chunk (level-1 never spills inside one)
constexpr int kStatusLostTicket = 1;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// gram.hip's split8, verbatim in effect: x = h + m + l, the three 8-bit fields of the 24-bit significand (truncation)
__device__ __forceinline__ void split8(const f32x4& lo4, const f32x4& hi4, u32x4& hp, u32x4& mp, u32x4& lp) {
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

// One workgroup: a 32-row block x 128 columns (8 steps).  Reads are 128-byte row segments, the transposition into
// fragment order goes through LDS, writes are whole 1 KiB fragment images.
constexpr int kSplitCols = 128;
__global__ __launch_bounds__(256) void plane_split_kernel(const float* __restrict__ G, int64_t n_rows, int64_t n_cols,
                                                          int64_t ld, const int32_t* __restrict__ row_index, int64_t k0,
                                                          int64_t n_steps, u32x4* __restrict__ planes) {
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
            const int64_t k = k0 + step0 * 16 + col;
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (k + 3 < n_cols) {
                v = *reinterpret_cast<const f32x4*>(src + k);
            } else {
                if (k + 0 < n_cols) v.x = src[k + 0];
                if (k + 1 < n_cols) v.y = src[k + 1];
                if (k + 2 < n_cols) v.z = src[k + 2];
            }
            *reinterpret_cast<f32x4*>(&tile[r][col]) = v;
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
        split8(lo4, hi4, h, m, l);
        u32x4* out = planes + ((rb * n_steps + step) * 3) * 64 + lane;
        out[0] = h;
        out[64] = m;
        out[128] = l;
    }
}

// SPS: 16-column steps per stage; NBUF: stages of LDS (the DMA runs NBUF - 1 stages ahead of the MFMAs).
// DBG (timing experiments only, wrong results): bit 0 no DMA after the first NBUF stages, bit 1 no MFMA.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {   // n is wave-uniform and at most 18
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        case 18: wait_vmcnt<18>(); break;
        default: wait_vmcnt<0>(); break;
    }
}

template <int SPS, int NBUF, int DBG>
__global__ __launch_bounds__(kThreads, 1) void gram_planes_kernel(const u32x4* __restrict__ planes, int64_t n_steps,
                                                                  int n_stages_total, double* __restrict__ partial,
                                                                  int n_tiles, const int2* __restrict__ tile_order,
                                                                  int n_chunks, int* __restrict__ tickets, int round_size,
                                                                  int t128, int slab_live0,
                                                                  int32_t* __restrict__ device_status) {
    constexpr int kRbBytes = SPS * 3 * kFragBytes;          // one 32-row block, one stage: [step][plane][1 KiB]
    constexpr int kStage = kRowBlocks * kRbBytes;           // 36,864 SPS
    constexpr int kPieces = kRowBlocks * SPS * 3;           // 1 KiB pieces per stage
    constexpr int kPerWave = (kPieces + 7) / 8;             // DMA instructions per wave and stage (the last may be idle)
    constexpr int kChunk = 512 / SPS;                       // stages per 8192-column chunk
    constexpr int kFlush = 16 / SPS;                        // stages per 256-column MFMA chain
    constexpr int kLead = NBUF - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [NBUF][12 row blocks][SPS steps][3 planes][1 KiB]

    // workgroup -> (tile, chunk): XCD x owns a contiguous share of the tile list and works through it chunk by chunk
    const int xcd = blockIdx.x & 7;
    const int seq = blockIdx.x >> 3;
    const int base = n_tiles >> 3, rem = n_tiles & 7;
    const int mine = base + (xcd < rem ? 1 : 0);
    const int first = xcd * base + (xcd < rem ? xcd : rem);
    const int chunk = mine > 0 ? seq / mine : n_chunks;
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
        if (chunk >= n_chunks) {
            if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    const int t_list = first + (seq - chunk * mine);
    const int2 tt = tile_order[t_list];
    const int bi = tt.x;   // 256-row block of the A side
    const int tj = tt.y;   // 128-row block of the B side

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ti = 2 * bi + (wr >> 1);                 // this wave's slab row
    const bool live_wave = tj <= ti && ti < t128;      // the upper half of a tile that straddles the diagonal is not needed

    const int stage0 = chunk * kChunk;
    int n_stages = n_stages_total - stage0;
    if (n_stages > kChunk) n_stages = kChunk;

    // LDS-DMA: piece q = wave + 8 i of the stage (row block q / (3 SPS), [step][plane] q % (3 SPS)); the LDS image of a
    // stage is piece-linear, and so is a row block's stage in HBM
    const u32x4* src[kPerWave];
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        int q = wave + 8 * i;
        if (q >= kPieces) q = kPieces - 1;
        const int rbl = q / (3 * SPS), piece = q % (3 * SPS);
        const int64_t rb = rbl < 8 ? static_cast<int64_t>(bi) * 8 + rbl : static_cast<int64_t>(tj) * 4 + (rbl - 8);
        src[i] = planes + ((rb * n_steps + static_cast<int64_t>(stage0) * SPS) * 3 + piece) * 64 + lane;
    }
    const int my_dmas = kPieces % 8 == 0 ? kPerWave : (wave < kPieces % 8 ? kPerWave : kPerWave - 1);
    auto dma = [&](int s, int buf) __attribute__((always_inline)) {
        if ((DBG & 1) && s >= NBUF) return;
        unsigned char* dst = lds + buf * kStage + wave * kFragBytes;
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            if (kPieces % 8 != 0 && i == kPerWave - 1 && wave >= kPieces % 8) break;   // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + static_cast<int64_t>(s) * (3 * SPS * 64)),
                                             (__attribute__((address_space(3))) void*)(dst + i * 8 * kFragBytes), 16, 0, 0);
        }
    };

    f32x16 acc[2][2], acc2[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[m][n][e] = 0.0f;
                acc2[m][n][e] = 0.0f;
            }

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* A = lds + buf * kStage + (2 * wr) * kRbBytes + lane * 16;
        const unsigned char* B = lds + buf * kStage + (8 + 2 * wc) * kRbBytes + lane * 16;
#pragma unroll
        for (int j = 0; j < SPS; ++j) {
            bf16x8 ap[3][2], bp[3][2];   // [plane h, m, l][block]
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ap[p][m] = *reinterpret_cast<const bf16x8*>(A + m * kRbBytes + (j * 3 + p) * kFragBytes);
                    bp[p][m] = *reinterpret_cast<const bf16x8*>(B + m * kRbBytes + (j * 3 + p) * kFragBytes);
                }
            // h h' + h m' + m h' + m m' + h l' + l h', smallest terms first, term-major over the four accumulators:
            // the order of gram.hip's split mode
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0};
            constexpr int pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[pa[t]][m], bp[pb[t]][n], acc[m][n], 0, 0, 0);
        }
    };
    auto flush = [&](int s) __attribute__((always_inline)) {
        if ((s + 1) % kFlush == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc2[m][n][e] += acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
        }
    };

    // the DMA runs kLead stages ahead; memory reads return in order, so "at most k stages' worth of my requests still
    // outstanding" is a vmcnt bound
    for (int a = 0; a < kLead && a < n_stages; ++a) dma(a, a);
    wait_vmcnt_dyn(((n_stages < kLead ? n_stages : kLead) - 1) * my_dmas);
    asm volatile("s_barrier" ::: "memory");
    int s = 0;
    for (; s + kLead < n_stages; ++s) {   // steady state: one stage issued, one consumed
        dma(s + kLead, (s + kLead) % NBUF);   // into the buffer the barrier of stage s - 1 freed
        if (live_wave && !(DBG & 2)) compute(s % NBUF);
        // stage s + 1 must have landed before anyone reads it; the kLead - 1 stages after it may still be in flight
        if (kPieces % 8 == 0 || wave < kPieces % 8) wait_vmcnt<(kLead - 1) * kPerWave>();
        else wait_vmcnt<(kLead - 1) * (kPerWave - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (live_wave) flush(s);
    }
    for (; s < n_stages; ++s) {           // drain: nothing left to issue
        if (live_wave && !(DBG & 2)) compute(s % NBUF);
        const int ahead = n_stages - 1 - (s + 1);
        wait_vmcnt_dyn((ahead > 0 ? ahead : 0) * my_dmas);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (live_wave) flush(s);
    }

    // epilogue: chunks of a tile add into its slabs in chunk order (gram.hip's ticket protocol)
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
        if (tid == 0) atomicOr(device_status, kStatusLostTicket);   // the host turns the word into BYZ_E_HIP
    } else if (live_wave) {
        const bool slab_live = chunk > 0 || slab_live0 != 0;
        double* out = partial + (static_cast<int64_t>(ti) * (ti + 1) / 2 + tj) * (kSlab * kSlab);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (wr & 1) * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const int j = wc * 64 + n * 32 + (lane & 31);
                    double v = static_cast<double>(acc2[m][n][e]);
                    v += static_cast<double>(acc[m][n][e]);
                    if (slab_live) v += out[i * kSlab + j];
                    out[i * kSlab + j] = v;
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

// The same tile, software-pipelined across the barrier: the MFMAs of stage s run from REGISTERS while the fragments of
// stage s + 1 are read from LDS, so nobody waits for an LDS read with the matrix pipe idle.  16-column stages, four LDS
// buffers, the DMA four stages ahead: the barrier that ends stage s comes after the reads of stage s + 1 (the last
// readers of that buffer) and after the wave's own pieces of stage s + 2 have landed.
template <int DBG>
__global__ __launch_bounds__(kThreads, 1) void gram_planes_pipe_kernel(const u32x4* __restrict__ planes, int64_t n_steps,
                                                                       int n_stages_total, double* __restrict__ partial,
                                                                       int n_tiles, const int2* __restrict__ tile_order,
                                                                       int n_chunks, int* __restrict__ tickets,
                                                                       int round_size, int t128, int slab_live0,
                                                                       int32_t* __restrict__ device_status) {
    constexpr int NBUF = 4;
    constexpr int kRbBytes = 3 * kFragBytes;
    constexpr int kStage = kRowBlocks * kRbBytes;           // 36,864
    constexpr int kPieces = kRowBlocks * 3;                 // 36
    constexpr int kPerWave = 5;                             // waves 0..3 move five pieces, waves 4..7 four
    constexpr int kChunk = 512;
    constexpr int kFlush = 16;
    constexpr int kLead = NBUF;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int xcd = blockIdx.x & 7;
    const int seq = blockIdx.x >> 3;
    const int base = n_tiles >> 3, rem = n_tiles & 7;
    const int mine = base + (xcd < rem ? 1 : 0);
    const int first = xcd * base + (xcd < rem ? xcd : rem);
    const int chunk = mine > 0 ? seq / mine : n_chunks;
    int* done = tickets + n_tiles + xcd;
    {
        const int round = round_size > 0 ? seq / round_size : 0;
        if (round > 0 && threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * round_size) {
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1u << 20)) break;
            }
        }
        __syncthreads();
        if (chunk >= n_chunks) {
            if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    const int t_list = first + (seq - chunk * mine);
    const int2 tt = tile_order[t_list];
    const int bi = __builtin_amdgcn_readfirstlane(tt.x);
    const int tj = __builtin_amdgcn_readfirstlane(tt.y);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int ti = 2 * bi + (wr >> 1);
    const bool live_wave = tj <= ti && ti < t128;

    const int stage0 = chunk * kChunk;
    int n_stages = n_stages_total - stage0;
    if (n_stages > kChunk) n_stages = kChunk;

    // piece q = wave + 8 i: a wave-uniform byte offset from `planes` (scalar registers) plus 16 bytes per lane
    int64_t piece_off[kPerWave];
#pragma unroll
    for (int i = 0; i < kPerWave; ++i) {
        int q = wave + 8 * i;
        if (q >= kPieces) q = kPieces - 1;
        const int rbl = q / 3, piece = q % 3;
        const int64_t rb = rbl < 8 ? static_cast<int64_t>(bi) * 8 + rbl : static_cast<int64_t>(tj) * 4 + (rbl - 8);
        piece_off[i] = (((rb * n_steps + stage0) * 3 + piece) * 64) * 16;
    }
    const unsigned char* lane_base = reinterpret_cast<const unsigned char*>(planes) + lane * 16;
    const int my_dmas = wave < kPieces % 8 ? kPerWave : kPerWave - 1;
    auto dma = [&](int s) __attribute__((always_inline)) {
        if ((DBG & 1) && s >= NBUF) return;
        unsigned char* dst = lds + (s % NBUF) * kStage + wave * kFragBytes;
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            if (i == kPerWave - 1 && wave >= kPieces % 8) break;   // wave-uniform
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(lane_base + piece_off[i] + static_cast<int64_t>(s) * (3 * kFragBytes)),
                (__attribute__((address_space(3))) void*)(dst + i * 8 * kFragBytes), 16, 0, 0);
        }
    };

    f32x16 acc[2][2], acc2[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[m][n][e] = 0.0f;
                acc2[m][n][e] = 0.0f;
            }

    struct Frags {
        bf16x8 a[3][2], b[3][2];   // [plane h, m, l][block]
    };
    auto read_frags = [&](int s, Frags& f) __attribute__((always_inline)) {
        const unsigned char* A = lds + (s % NBUF) * kStage + (2 * wr) * kRbBytes + lane * 16;
        const unsigned char* B = lds + (s % NBUF) * kStage + (8 + 2 * wc) * kRbBytes + lane * 16;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f.a[p][m] = *reinterpret_cast<const bf16x8*>(A + m * kRbBytes + p * kFragBytes);
                f.b[p][m] = *reinterpret_cast<const bf16x8*>(B + m * kRbBytes + p * kFragBytes);
            }
    };
    auto multiply = [&](const Frags& f) __attribute__((always_inline)) {
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0};   // gram.hip's order: l h', h l', m m', m h', h m', h h'
        constexpr int pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[pa[t]][m], f.b[pb[t]][n], acc[m][n], 0, 0, 0);
    };
    auto flush = [&](int s) __attribute__((always_inline)) {
        if ((s + 1) % kFlush == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        acc2[m][n][e] += acc[m][n][e];
                        acc[m][n][e] = 0.0f;
                    }
        }
    };
    // one stage: multiply `cur` (stage s, in registers), read stage s + 1 into `next`, keep the DMA kLead stages ahead
    auto steady = [&](int s, const Frags& cur, Frags& next) __attribute__((always_inline)) {
        dma(s + kLead);                       // into the buffer of stage s, whose last readers passed the previous barrier
        if (live_wave && !(DBG & 2)) {
            read_frags(s + 1, next);
            multiply(cur);
        }
        // before anyone reads stage s + 2 it must have landed: of my requests only stages s + 3, s + 4 may be outstanding
        if (wave < kPieces % 8) wait_vmcnt<2 * kPerWave>();
        else wait_vmcnt<2 * (kPerWave - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (live_wave) flush(s);
    };
    auto drain = [&](int s, const Frags& cur, Frags& next) __attribute__((always_inline)) {
        if (s + kLead < n_stages) dma(s + kLead);
        if (live_wave && !(DBG & 2)) {
            if (s + 1 < n_stages) read_frags(s + 1, next);
            multiply(cur);
        }
        const int last = s + kLead < n_stages ? s + kLead : n_stages - 1;
        const int ahead = last - (s + 2);
        wait_vmcnt_dyn((ahead > 0 ? ahead : 0) * my_dmas);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (live_wave) flush(s);
    };

    Frags x, y;
    for (int a = 0; a < kLead && a < n_stages; ++a) dma(a);
    {
        const int issued = n_stages < kLead ? n_stages : kLead;
        wait_vmcnt_dyn((issued > 2 ? issued - 2 : 0) * my_dmas);   // stages 0 and 1 have landed
    }
    asm volatile("s_barrier" ::: "memory");
    if (live_wave && !(DBG & 2) && n_stages > 0) read_frags(0, x);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // everybody has read stage 0: its buffer may be refilled
    int s = 0;
    for (; s + 1 + kLead < n_stages; s += 2) {
        steady(s, x, y);
        steady(s + 1, y, x);
    }
    for (; s < n_stages; s += 2) {
        drain(s, x, y);
        if (s + 1 < n_stages) drain(s + 1, y, x);
    }

    // epilogue: chunks of a tile add into its slabs in chunk order (gram.hip's ticket protocol)
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
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = (wr & 1) * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const int j = wc * 64 + n * 32 + (lane & 31);
                    double v = static_cast<double>(acc2[m][n][e]);
                    v += static_cast<double>(acc[m][n][e]);
                    if (slab_live) v += out[i * kSlab + j];
                    out[i * kSlab + j] = v;
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

int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : fallback;
}

}  // namespace

bool gram_planes_enabled() { return env_int("BYZ_GRAM_PLANES", 1) != 0; }

// Fills `slabs` (gram.hip's fp64 slab format: one 128 x 128 slab per lower-triangle tile ti (ti + 1) / 2 + tj) with the
// Gram of the n_rows logical rows G[row_index[r]]; with share_count > 1 only this share's tiles (owned[] says which).
int launch_gram_planes(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                       double* slabs, int share_count, int share_index, uint8_t* owned_host, hipStream_t stream) {
    const int64_t t128 = ceil_div(n_rows, kSlab);
    const int64_t t256 = ceil_div(t128, 2);
    const int64_t rows_pad = t256 * kWgRows;
    // tile list: 8 x 8 super-blocks of slabs = 4 x 8 workgroup tiles, an XCD's resident workgroups share few row blocks
    if (ctx->plane_order_T != t128 || ctx->plane_order_share != share_count * 65536 + share_index) {
        ctx->plane_order_host.clear();
        const int64_t S = ceil_div(t128, 8);
        int64_t position = 0;
        for (int64_t I = 0; I < S; ++I)
            for (int64_t J = 0; J <= I; ++J)
                for (int64_t bi = I * 4; bi < I * 4 + 4 && bi < t256; ++bi)
                    for (int64_t tj = J * 8; tj < J * 8 + 8 && tj <= 2 * bi + 1 && tj < t128; ++tj) {
                        if (position++ % share_count != share_index) continue;
                        ctx->plane_order_host.push_back(static_cast<int32_t>(bi));
                        ctx->plane_order_host.push_back(static_cast<int32_t>(tj));
                    }
        BYZ_TRY(ctx->plane_order.ensure(ctx->plane_order_host.size() * sizeof(int32_t) + 16));
        BYZ_HIP(hipMemcpyAsync(ctx->plane_order.ptr, ctx->plane_order_host.data(),
                               ctx->plane_order_host.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        BYZ_HIP(hipStreamSynchronize(stream));
        ctx->plane_order_T = t128;
        ctx->plane_order_share = share_count * 65536 + share_index;
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
    const int64_t chunk_cols = static_cast<int64_t>(kChunkStages) * kStageCols;
    size_t free_b = 0, total_b = 0;
    BYZ_HIP(hipMemGetInfo(&free_b, &total_b));
    int64_t budget = static_cast<int64_t>(env_int("BYZ_GRAM_PLANE_MB", 16384)) << 20;
    const int64_t have = static_cast<int64_t>(ctx->gram_planes.bytes);
    if (budget > have + static_cast<int64_t>(free_b / 2)) budget = have + static_cast<int64_t>(free_b / 2);
    int64_t chunks_per_sc = budget / (rows_pad * 6 * chunk_cols);
    const int64_t chunks_total = ceil_div(n_cols, chunk_cols);
    if (chunks_per_sc > chunks_total) chunks_per_sc = chunks_total;
    if (chunks_per_sc < 1) {
        set_error("gram: no room for one 8192-column chunk of bf16 planes (%lld rows)", (long long)rows_pad);
        return BYZ_E_HIP;
    }
    // even super-chunks: the same number of launches, the last one not a stub
    const int64_t n_sc = ceil_div(chunks_total, chunks_per_sc);
    chunks_per_sc = ceil_div(chunks_total, n_sc);
    const int64_t sc_cols = chunks_per_sc * chunk_cols;
    BYZ_TRY(ctx->gram_planes.ensure(static_cast<size_t>(rows_pad) * 6 * sc_cols));
    BYZ_TRY(ctx->gram_tickets.ensure(static_cast<size_t>(n_tiles + 8) * sizeof(int)));
    int* tickets = ctx->gram_tickets.as<int>();
    u32x4* planes = ctx->gram_planes.as<u32x4>();
    // variant: 0 = 16-column stages, 4 LDS buffers (DMA three stages ahead); 1 = 32-column stages, 2 buffers;
    // 2 = 16-column stages, 3 buffers; 10 + v / 20 + v: timing experiments (no DMA / no MFMA, wrong results)
    const int variant = env_int("BYZ_GRAM_PLANES_VARIANT", 0);
    typedef void (*kernel_t)(const u32x4*, int64_t, int, double*, int, const int2*, int, int*, int, int, int, int32_t*);
    kernel_t kernel = &gram_planes_kernel<1, 4, 0>;
    int sps = 1, nbuf = 4;
    switch (variant) {
        case 1: kernel = &gram_planes_kernel<2, 2, 0>; sps = 2; nbuf = 2; break;
        case 2: kernel = &gram_planes_kernel<1, 3, 0>; sps = 1; nbuf = 3; break;
        case 3: kernel = &gram_planes_pipe_kernel<0>; break;
        case 13: kernel = &gram_planes_pipe_kernel<1>; break;
        case 23: kernel = &gram_planes_pipe_kernel<2>; break;
        case 10: kernel = &gram_planes_kernel<1, 4, 1>; break;
        case 20: kernel = &gram_planes_kernel<1, 4, 2>; break;
        case 11: kernel = &gram_planes_kernel<2, 2, 1>; sps = 2; nbuf = 2; break;
        case 21: kernel = &gram_planes_kernel<2, 2, 2>; sps = 2; nbuf = 2; break;
        default: break;
    }
    const size_t lds_bytes = static_cast<size_t>(nbuf) * sps * kRowBlocks * 3 * kFragBytes;
    BYZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lds_bytes)));
    const int round_size = env_int("BYZ_GRAM_ROUND", ctx->num_cus / 8);   // one workgroup per CU
    const int64_t per_xcd = ceil_div(n_tiles, 8);
    for (int64_t sc = 0; sc < n_sc; ++sc) {
        const int64_t k0 = sc * sc_cols;
        const int64_t cols = n_cols - k0 < sc_cols ? n_cols - k0 : sc_cols;
        const int64_t stages = ceil_div(cols, kStageCols);
        const int64_t n_steps = stages * 2;
        const int64_t n_chunks = ceil_div(stages, kChunkStages);
        {
            KernelTimer t(ctx, BYZ_K_PLANE_SPLIT, stream);
            const dim3 grid(static_cast<unsigned>(ceil_div(n_steps, kSplitCols / 16)), static_cast<unsigned>(rows_pad / 32));
            plane_split_kernel<<<grid, 256, 0, stream>>>(G, n_rows, n_cols, ld, row_index, k0, n_steps, planes);
            BYZ_TRY(check_launch("plane_split_kernel"));
        }
        BYZ_HIP(hipMemsetAsync(tickets, 0, static_cast<size_t>(n_tiles + 8) * sizeof(int), stream));
        {
            KernelTimer t(ctx, BYZ_K_GRAM, stream);
            const int64_t grid = 8 * per_xcd * n_chunks;
            if (grid > 0x7fffffff) {
                set_error("gram: grid too large");
                return BYZ_E_UNSUPPORTED;
            }
            kernel<<<static_cast<unsigned>(grid), kThreads, lds_bytes, stream>>>(
                planes, n_steps, static_cast<int>(n_steps / sps), slabs, static_cast<int>(n_tiles), ctx->plane_order.as<int2>(),
                static_cast<int>(n_chunks), tickets, round_size, static_cast<int>(t128), sc > 0 ? 1 : 0,
                device_status_word(ctx));
            BYZ_TRY(check_launch("gram_planes_kernel"));
        }
    }
    return BYZ_OK;
}

}  // namespace byz
