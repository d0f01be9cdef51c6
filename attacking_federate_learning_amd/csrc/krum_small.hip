// Krum for the reference's own sizes, N <= 128 clients (defences.py:16-42): TWO launches (round 2: five; the general path: ~15).
//
// The general path (gram.hip + select.hip) is built for N in the thousands: at N = 100, D = 79,510 (BASELINE configs[1]) it
// spends 69 us per round in ~15 kernel launches and memsets.  Round 2's five launches took 30 us: K1 14.7 us, and 15 us for
// four trivial dependent kernels (measured one by one, profiles/r04c_c2_launch_costs.txt: reduce 6.1, distances 1.5, score
// 3.6, pick 3.7 -- a dependent launch costs ~3.6 us whatever it does, and the reduce walked its slabs in dependent round
// trips).  Round 4: everything behind the Gram is ONE kernel, because everything a row's Krum score needs is that row.
//
//   K1 small_gram_kernel     all N rows x a 128-column slice per step, one row of the slice per thread: global fp32 ->
//                            registers -> (row, slice) power-of-two scale -> two fp16 planes in LDS (row-major, 272-byte
//                            pitch: conflict-free ds_write_b64 and ds_read_b128) -> v_mfma_f32_32x32x16_f16, three per
//                            32 x 32 block and 16 columns -- (sum m h' + sum h m') + sum h h' on three accumulators, a
//                            SYMMETRIC function of the two rows (gram_planes.hip's f16x2 planes, 6e-8 against fp64) --
//                            lower-triangle blocks only, spread over the 8 waves so that no SIMD carries more than 3; the
//                            next two slices' loads are in flight meanwhile.  One full 128 x 128 fp32 slab (both
//                            orientations of every block) and one compact diagonal per workgroup.
//   K2 small_rows_kernel     one workgroup per row i: the row of the Gram and the diagonal summed over the slabs in fp64
//                            (fixed order, every load in flight at once), d_ij = sqrt(max(0, c_ii + c_jj - 2 c_ij)); pairs
//                            the identity cannot resolve (d^2 < (c_ii + c_jj) / 16) re-evaluated on the difference itself
//                            (defences.py:20), identical rows folded to one proof each; in-register bitonic sort of the
//                            row's distances by one wave and the SEQUENTIAL fp32 sum of the first n - f, exactly as
//                            Python's sum() forms it (defences.py:33-34); a ticket, the last workgroup finds the winner
//                            (visit order 1, 0, 2, ..., strict '<' against 1e20, defences.py:27-37), everybody copies its
//                            share of the winning row.
//   K4 small_score_kernel, K5 small_pick_kernel: the selection on a distance matrix the CALLER supplies
//                            (`krum(..., distances=...)`, byz_krum_select_dev): score per row, argmin + row copy.
//
// Algorithmic traffic: 4 N D bytes read once (K1); everything else is O(N^2).  Bound: HBM (N / 4 flop per byte is below
// the machine balance of the 16-bit matrix pipe for every N <= 128); in fact cache latency and the one kernel boundary.
#include "common.hpp"
#include "lane_exchange.hpp"

#include <cstdio>
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
constexpr int kSlabFloats = kMaxRows * kMaxRows;    // a workgroup's partial Gram: full 128 x 128, both triangles
constexpr int kPairChunk = 8192;                    // columns per (pair, chunk) work item of the near-duplicate pass
constexpr double kNearEps = 1.0 / 16.0;             // gram.hip's threshold
constexpr float kKrumInit = 1e20f;                  // defences.py:27
constexpr int kStatusFalseTwin = 4;                 // bits of the context's sticky status word (gram.hip)
constexpr int kStatusSmallTimeout = 8;
constexpr unsigned kSpinLimit = 1u << 18;   // ~0.1-0.3 s: the worker publishes within microseconds

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

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
__global__ __launch_bounds__(kThreads, 1) void small_gram_kernel(const float* __restrict__ G, int n_rows, int64_t n_cols,
                                                                 int64_t ld, int n_slices, float* __restrict__ slabs,
                                                                 float* __restrict__ diag_slabs, float* __restrict__ score_board,
                                                                 int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, q4 = lane & 31;
    const int n_rb = (n_rows + 31) >> 5;                 // live 32-row blocks
    // K2's arrival board: the previous call's K2 is done with it, this call's K2 starts behind this kernel
    if (blockIdx.x == 0 && tid < kMaxRows) reinterpret_cast<uint32_t*>(score_board)[tid] = 0xffc0deadu;

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

    // Three MFMA chains per block -- x = sum m_a h_b, y = sum h_a m_b, z = sum h_a h_b -- combined as (x + y) + z.  Swapping the
    // two operands swaps x and y and leaves every product and every chain as it is, and fp32 addition commutes: the Gram
    // entry is a SYMMETRIC function of the two rows' planes, c(a, b) == c(b, a) bit for bit.  (One chain m h' + h m' + h h'
    // added the cross terms in an order that depended on which operand a row was.)  Identical rows therefore get bitwise
    // identical Gram ROWS -- not just c_ii == c_ij == c_jj -- and identical distance rows fall out of the arithmetic; the
    // canonicalisation pass the first small path needed (a workgroup that rewrote the whole matrix) is gone.
    f32x16 accx[2], accy[2], accz[2], sum[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            accx[b][e] = 0.0f;
            accy[b][e] = 0.0f;
            accz[b][e] = 0.0f;
            sum[b][e] = 0.0f;
        }

    auto multiply = [&](auto nb_c, const unsigned char* buf) __attribute__((always_inline)) {
        constexpr int NB = decltype(nb_c)::value;
        const int* shifts = reinterpret_cast<const int*>(buf + 2 * kPlaneBytes);
        const unsigned char* a0 = buf + (32 * bi0 + q4) * kPitch + half * 16;
        const unsigned char* b0 = buf + (32 * bj0 + q4) * kPitch + half * 16;
        const unsigned char* a1 = buf + (32 * bi1 + q4) * kPitch + half * 16;
        const unsigned char* b1 = buf + (32 * bj1 + q4) * kPitch + half * 16;
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
            accx[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, accx[0], 0, 0, 0);
            if constexpr (NB == 2) accx[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cm, dh, accx[1], 0, 0, 0);
            accy[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, accy[0], 0, 0, 0);
            if constexpr (NB == 2) accy[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, dm, accy[1], 0, 0, 0);
            accz[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, accz[0], 0, 0, 0);
            if constexpr (NB == 2) accz[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, dh, accz[1], 0, 0, 0);
        }
        // undo the rows' scales (powers of two: exact) and add the slice to the running sums
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int bi = b == 0 ? bi0 : bi1, bj = b == 0 ? bj0 : bj1;
            const int sj = shifts[32 * bj + q4];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int si = shifts[32 * bi + (e & 3) + 8 * (e >> 2) + 4 * half];
                sum[b][e] += __builtin_ldexpf((accx[b][e] + accy[b][e]) + accz[b][e], -(si + sj));
                accx[b][e] = 0.0f;
                accy[b][e] = 0.0f;
                accz[b][e] = 0.0f;
            }
        }
    };

    // scale + split of one slice into the LDS buffer `buf` (planes, then the rows' shifts)
    // MASKED: the slice may be the ragged one (its window overlaps the slice before it) or lie past the end (all zeros)
    auto split_to = [&](f32x4 (&v)[8], int s, unsigned char* buf, auto masked_c) __attribute__((always_inline)) {
        constexpr bool kMasked = decltype(masked_c)::value;
        int* shifts = reinterpret_cast<int*>(buf + 2 * kPlaneBytes);
        if (dbg & 2) {           // timing experiment: the loads are consumed, nothing is split
            float any = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) any += v[k][0];
            if (any == 1.2345e38f) shifts[r_local] = 1;
            return;
        }
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
        unsigned char* dst = buf + r_local * kPitch + chunk * 8;
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
    };
    auto multiply_from = [&](const unsigned char* buf) __attribute__((always_inline)) {
        if (dbg & 1) return;     // timing experiment: no MFMAs (BYZ_KRUM_SMALL_DBG)
        if (nb == 2) multiply(std::integral_constant<int, 2>{}, buf);
        else if (nb == 1) multiply(std::integral_constant<int, 1>{}, buf);
    };
    // one slice through ONE buffer (the loop form for long rows): split, MFMAs, two barriers
    auto consume = [&](f32x4 (&v)[8], int s, int next_slice, auto prefetch_c, auto masked_c) __attribute__((always_inline)) {
        constexpr bool kPrefetch = decltype(prefetch_c)::value;
        split_to(v, s, lds, masked_c);
        __syncthreads();
        if constexpr (!TINY && kPrefetch) load_slice(v, next_slice);   // lands while the MFMAs run (a dummy word past the last slice)
        multiply_from(lds);
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
        // the host sizes the grid so that (PER - 1) * grid < n_slices: only the last step can hold the ragged slice or none.
        // (Round 4 measured a double-buffered form -- two LDS buffers, one barrier per slice, waves 0-3 splitting slice c + 1
        // while waves 4-7 multiply slice c -- at 16.1 us against 15.6-16.0 for this one: the kernel's time is additive in its
        // parts -- 9.4 us of loads + stores + launch, 5.0 of MFMAs, 1.6 of splitting, profiles/r04j_k1_decomposition.txt --
        // because with three slices per workgroup the "pipeline" is all fill and drain.  Removed again.)
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

    // The workgroup's slab: a full 128 x 128 row-major matrix (pitch kMaxRows), BOTH orientations of every off-diagonal
    // block, so that whoever owns row i in K2 finds c_i0 .. c_i,127 as 512 contiguous bytes per slab.  The mirror image of
    // an off-diagonal block goes out as 16-byte stores (four consecutive rows of the block = four consecutive columns of
    // its transpose sit in consecutive accumulator registers).
    float* out = slabs + static_cast<int64_t>(blockIdx.x) * kSlabFloats;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b < nb) {
            const int bi = b == 0 ? bi0 : bi1, bj = b == 0 ? bj0 : bj1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int il = (e & 3) + 8 * (e >> 2) + 4 * half;
                out[(32 * bi + il) * kMaxRows + 32 * bj + q4] = sum[b][e];
                // the diagonal once more, compactly: K2 needs c_jj of every row next to its own row of the Gram
                if (bi == bj && il == q4) diag_slabs[static_cast<int64_t>(blockIdx.x) * kMaxRows + 32 * bi + q4] = sum[b][e];
            }
            if (bi != bj) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = sum[b][4 * e4 + e];
                    *reinterpret_cast<f32x4*>(out + (32 * bj + q4) * kMaxRows + 32 * bi + 8 * e4 + 4 * half) = v;
                }
            }
        }
    }
}

// ---- K2 ---------------------------------------------------------------------------------------------------------------
// One workgroup per ROW of the distance matrix (round 4; it replaces the reduce kernel, the distance kernel with its worker /
// helper hand-off, the score kernel and the pick kernel of round 2: five launches -> two).  Everything a row's Krum score
// needs is the row itself, so nothing here waits for another workgroup until the very end:
//
//   1. c_i0 .. c_i,127 and the diagonal c_00 .. c_127,127 = sums over the K1 workgroups' slabs, fp64, in ONE fixed order for
//      every entry (16 groups of threads take the slabs g, g + 16, ...; a fixed tree joins the groups).  Every load of the
//      workgroup is in flight at once -- 16-byte loads, all issued before the first add; the reduce kernel of round 2 walked
//      the slabs in dependent round trips and took 6 us of a 30 us round (profiles/r04c_c2_launch_costs.txt).
//      c_ij here and c_ji in row j's workgroup are sums of the same numbers in the same order (K1 stores both orientations,
//      and its arithmetic is symmetric): the matrix comes out symmetric bit for bit, and identical rows get identical rows.
//   2. d_ij = sqrt(max(0, c_ii + c_jj - 2 c_ij)); a pair the Gram identity cannot resolve (d^2 < (c_ii + c_jj) / 16, or an
//      exact zero) is re-evaluated on the difference itself (defences.py:20) by THIS workgroup -- and by row j's, which runs
//      the same code on the same bytes and gets the same bits.  Identical rows (bitwise equal c_ii, c_ij, c_jj: the attack's
//      clients, malicious.py:26-27) are folded first: only the proof pair (i, first twin) is evaluated, and it must come out
//      as zero (a refuted nomination sets the status word, as in gram.hip).
//   3. the row's n - 1 distances sorted in registers by one wave, the sequential fp32 sum of the first n - f (defences.py:33-34).
//   4. a ticket; the LAST workgroup to arrive runs the argmin over the scores (visit order 1, 0, 2, ...; strict '<' against
//      1e20) and publishes the winner; the others wait for it (all n <= 128 workgroups are resident: bounded spin, time-out
//      -> status word) and everybody copies its share of the winning row.
struct RowsArgs {
    const float* slabs;        // [n_slabs][128][128] fp32
    const float* diag_slabs;   // [n_slabs][128]
    int n_slabs;
    int n;                     // rows
    const float* G;
    int64_t n_cols, ld;
    float* dist;               // n x n, pitch n
    int prefix_len;            // < 0: distances only
    float* scores;             // n
    int32_t* winner;           // device word the index goes to
    float* out_row;            // optional: copy of the winning row
    int32_t* status;
};

__device__ __forceinline__ unsigned long long bits_of(double v) { return static_cast<unsigned long long>(__double_as_longlong(v)); }

// sum over all columns of (a - b)^2: the difference in fp32 as the reference forms it (defences.py:20), squares and sums in
// fp64; chunks of kPairChunk columns in order, inside a chunk a fixed partition over the 512 threads and a fixed tree -- the
// result depends on the two rows' bytes only, not on who computes it or which of the two comes first ((a - b)^2 == (b - a)^2).
__device__ double pair_sq_distance(const float* __restrict__ a, const float* __restrict__ b, int64_t n_cols, double* red) {
    double total = 0.0;
    for (int64_t k0 = 0; k0 < n_cols; k0 += kPairChunk) {
        const int64_t k1 = k0 + kPairChunk < n_cols ? k0 + kPairChunk : n_cols;
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
        total += red[0];
        __syncthreads();
    }
    return total;
}

// development aid (BYZ_KRUM_SMALL_TIMING=1): s_memtime stamps of K2's phases, taken by thread 0 of every workgroup
constexpr int kRowStamps = 8;
__device__ unsigned long long g_rows_stamps[kMaxRows * kRowStamps];
#define BYZ_STAMP(k)                                                                              \
    do {                                                                                          \
        if (timing && threadIdx.x == 0) g_rows_stamps[blockIdx.x * kRowStamps + (k)] = __builtin_amdgcn_s_memtime();   \
    } while (0)

constexpr uint32_t kScoreSentinel = 0xffc0dead;   // "no score yet": K1 writes it, K2's rows overwrite it (never a published score)
constexpr int kGroups = 16;                       // thread groups of the slab sums (32 threads x 4 columns each)
constexpr int kMaxPerGroup = 256 / kGroups;       // K1 launches at most 256 workgroups: at most 16 slabs per group

// TIMING (BYZ_KRUM_SMALL_TIMING): the phase stamps.  A template parameter: as a flag in device memory it was a dependent scalar
// load at the head of a 10 us kernel.
// PG: slabs per thread group, a multiple of 4 with 16 PG >= n_slabs (the loads of a group are all issued before its first add:
// with fewer slabs than 256 the shorter forms issue fewer of them).
template <bool TIMING, int PG>
__global__ __launch_bounds__(kThreads, 1) void small_rows_kernel(RowsArgs p) {
    static_assert(PG % 4 == 0 && PG >= 4 && PG <= kMaxPerGroup, "slabs per group");
    __shared__ double part[2][kGroups][kMaxRows];   // [row entries | diagonal][group][column]
    __shared__ double red[kThreads];
    __shared__ double c_row[kMaxRows], c_diag[kMaxRows];
    __shared__ float d_row[kMaxRows];
    __shared__ unsigned char what[kMaxRows];        // 0 nothing, 1 re-evaluate on the difference, 2 twin of this row
    __shared__ int words[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x;
    const int n = p.n;
    constexpr bool timing = TIMING;
    BYZ_STAMP(0);

    // ---- 1. the row of the Gram and the diagonal
    {
        const int jq = tid & 31, grp = tid >> 5;
        f32x4 rv[PG], dv[PG];
        const int jq_read = 4 * jq < n ? jq : 0;     // columns past n: the lane repeats lane 0's request (no extra bytes)
        const float* row_src = p.slabs + static_cast<int64_t>(i) * kMaxRows + 4 * jq_read;
        const float* diag_src = p.diag_slabs + 4 * jq_read;
#pragma unroll
        for (int k = 0; k < PG; ++k) {
            const int g = grp + kGroups * k;
            const int gg = g < p.n_slabs ? g : 0;      // a clamped, unconditional load; discarded below
            rv[k] = *reinterpret_cast<const f32x4*>(row_src + static_cast<int64_t>(gg) * kSlabFloats);
            dv[k] = *reinterpret_cast<const f32x4*>(diag_src + static_cast<int64_t>(gg) * kMaxRows);
        }
        // Four slabs at a time are first added in fp32 -- (a + b) + (c + d), the same kind of rounding K1 makes when it adds
        // its slices into a slab, on numbers a fiftieth of the total -- and the quads in fp64: a quarter of the conversions
        // and fp64 additions, which (with the load issue) were what this phase cost (9,300 of the kernel's 24,000 ticks).
        // Slabs past the last are zeros.  The order is the same for every entry of every row: symmetry and ties are kept.
        double rs[4] = {0.0, 0.0, 0.0, 0.0}, ds[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < PG; ++k) {
            if (!(grp + kGroups * k < p.n_slabs)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rv[k][e] = 0.0f;
                    dv[k][e] = 0.0f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PG; k += 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                rs[e] += static_cast<double>(__fadd_rn(__fadd_rn(rv[k][e], rv[k + 1][e]), __fadd_rn(rv[k + 2][e], rv[k + 3][e])));
                ds[e] += static_cast<double>(__fadd_rn(__fadd_rn(dv[k][e], dv[k + 1][e]), __fadd_rn(dv[k + 2][e], dv[k + 3][e])));
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            part[0][grp][4 * jq + e] = rs[e];
            part[1][grp][4 * jq + e] = ds[e];
        }
    }
    __syncthreads();
    if (tid < 2 * kMaxRows) {
        const int v = tid >> 7, j = tid & (kMaxRows - 1);
        double t8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t8[k] = part[v][2 * k][j] + part[v][2 * k + 1][j];
        const double tot = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
        if (v == 0) c_row[j] = tot;
        else c_diag[j] = tot;
    }
    if (tid == 0) words[0] = 0;
    __syncthreads();
    BYZ_STAMP(1);

    // ---- 2. distances; what the Gram identity cannot resolve
    if (tid < kMaxRows) {
        const int j = tid;
        float d = __builtin_inff();      // the self slot and the padding: the reference keeps no self-distance (defences.py:18-20)
        unsigned char w = 0;
        if (j < n && j != i) {
            // c_ii and c_jj are the diagonal's sums, c_ij this row's: for identical rows all three are the same bits
            const double cii = c_diag[i], cjj = c_diag[j], cij = c_row[j];
            const double d2 = cii + cjj - 2.0 * cij;
            d = static_cast<float>(sqrt(d2 < 0.0 ? 0.0 : d2));   // NaN (poisoned input) stays NaN
            if (d2 < kNearEps * (cii + cjj) || d == 0.0f) {     // (both false for NaN)
                const bool twin = bits_of(cij) == bits_of(cii) && bits_of(cjj) == bits_of(cii);
                w = twin ? 2 : 1;
            }
        }
        d_row[j] = d;
        what[j] = w;
        if (w != 0) atomicOr(&words[0], w);
    }
    __syncthreads();
    if (words[0] != 0) {   // uniform; never taken for clients that are neither identical nor nearly so
        const float* mine = p.G + static_cast<int64_t>(i) * p.ld;
        bool proven = false;
        for (int j = 0; j < n; ++j) {
            const int w = what[j];          // uniform
            if (w == 0) continue;
            if (w == 2) {
                // identical rows: ONE proof per row -- against the first member of its class, if that is not this row
                if (proven || j > i) continue;
                proven = true;
                const double sq = pair_sq_distance(mine, p.G + static_cast<int64_t>(j) * p.ld, p.n_cols, red);
                if (sq != 0.0 && tid == 0) atomicOr(p.status, kStatusFalseTwin);
                continue;
            }
            const double sq = pair_sq_distance(mine, p.G + static_cast<int64_t>(j) * p.ld, p.n_cols, red);
            if (tid == 0) d_row[j] = static_cast<float>(sqrt(sq));
        }
        __syncthreads();
    }
    if (tid < n) p.dist[static_cast<int64_t>(i) * n + tid] = d_row[tid];
    BYZ_STAMP(2);
    if (p.prefix_len < 0) return;

    // ---- 3. the score: one wave sorts the row in registers (two values per lane, bitonic network); the prefix is then added
    // left to right out of the registers (v_readlane with a uniform lane index: ~8 cycles per term where a walk through LDS
    // took ~70).  (Sorting by counting on all eight waves -- every thread a quarter of the row's compares -- was measured:
    // 8,600 ticks against 4,100 for the network.)
    if (wave == 0) {
        float x[1][2];
#pragma unroll
        for (int r = 0; r < 2; ++r) x[0][r] = d_row[r + 2 * lane];     // +inf in the self slot and past n
        lanes::wave_bitonic_sort<2, 1>(x, lane);
        const float x0 = x[0][0], x1 = x[0][1];                        // sorted index 2 l, 2 l + 1 live in lane l
        float sc = 0.0f;
        const int pairs = p.prefix_len >> 1;
        for (int l = 0; l < pairs; ++l) {
            sc = __fadd_rn(sc, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), l)));
            sc = __fadd_rn(sc, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x1), l)));
        }
        if (p.prefix_len & 1)
            sc = __fadd_rn(sc, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x0), pairs)));
        // ---- 4. the score goes out as ONE relaxed device-scope store and doubles as this row's arrival: K1 filled the score
        // array with a sentinel (a NaN no score can be: a NaN score is published as the canonical quiet NaN), nothing else
        // has to be visible to the other workgroups, so there is no fence and no counter
        if (sc != sc) sc = __uint_as_float(0x7fc00000u);
        if (lane == 0) __hip_atomic_store(p.scores + i, sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    BYZ_STAMP(3);
    if (p.out_row == nullptr && i != 0) return;     // only the index is wanted: workgroup 0 reports it
    // Everybody who needs the winner polls the n scores until none is the sentinel and runs the argmin on what it polled
    // (128 scores, two per lane).  All n <= 128 workgroups are resident: the spin is bounded, a time-out sets the status word.
    if (wave == 0) {
        float s0 = 0.0f, s1 = 0.0f;
        unsigned spins = 0;
        bool ok = true;
        while (true) {
            const uint32_t b0 = lane < n ? __hip_atomic_load(reinterpret_cast<const uint32_t*>(p.scores) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const uint32_t b1 = lane + 64 < n ? __hip_atomic_load(reinterpret_cast<const uint32_t*>(p.scores) + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            s0 = __uint_as_float(b0);
            s1 = __uint_as_float(b1);
            if (__ballot(b0 == kScoreSentinel || b1 == kScoreSentinel) == 0ull) break;
            if (++spins > kSpinLimit) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok && lane == 0) atomicOr(p.status, kStatusSmallTimeout);
        if (timing && lane == 0) g_rows_stamps[blockIdx.x * kRowStamps + 4] = __builtin_amdgcn_s_memtime();
        // argmin in visit order 1, 0, 2, ... with a strict '<' against 1e20 (defences.py:27-37) = the smallest 64-bit key
        // (order-preserving score bits << 32 | visit position) among the scores below 1e20 (NaN and +inf are not; + 0.0f
        // folds a -0.0 onto +0.0, which compare equal)
        unsigned long long key = ~0ull;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int u = lane + 64 * r;
            const float sc = (r == 0 ? s0 : s1) + 0.0f;
            if (u < n && sc < kKrumInit) {     // false for NaN, as in the reference
                const uint32_t bits = __float_as_uint(sc);
                const uint32_t ordered = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);   // monotone in the value
                const unsigned long long k = (static_cast<unsigned long long>(ordered) << 32) | static_cast<unsigned>(visit_position(u));
                key = k < key ? k : key;
            }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const uint32_t lo = static_cast<uint32_t>(key), hi = static_cast<uint32_t>(key >> 32);
            const uint32_t olo = __float_as_uint(lanes::lane_xor(__uint_as_float(lo), m, lane));
            const uint32_t ohi = __float_as_uint(lanes::lane_xor(__uint_as_float(hi), m, lane));
            const unsigned long long other = (static_cast<unsigned long long>(ohi) << 32) | olo;
            key = other < key ? other : key;
        }
        int row = -1;
        if (key != ~0ull) {
            const int vp = static_cast<int>(key & 0xffffffffu);
            row = vp == 0 ? 1 : (vp == 1 ? 0 : vp);
        }
        if (lane == 0) {
            // a single row has an empty distance dict in the reference: nothing is visited, the index stays -1
            int chosen = n < 2 ? -1 : row;
            if (i == 0 && ok) *p.winner = chosen;
            if (chosen < 0) chosen += n;      // numpy's G[-1]: the reference returns the last row when nothing won
            words[2] = ok ? chosen : -2;
        }
    }
    __syncthreads();
    BYZ_STAMP(5);
    if (p.out_row == nullptr || words[2] < 0) return;
    // ---- everybody copies its share of the winning row
    const float* src = p.G + static_cast<int64_t>(words[2]) * p.ld;
    const int64_t per = ((p.n_cols + gridDim.x - 1) / gridDim.x + 3) & ~static_cast<int64_t>(3);
    const int64_t k0 = per * blockIdx.x;
    const int64_t k1 = k0 + per < p.n_cols ? k0 + per : p.n_cols;
    for (int64_t k = k0 + tid; k < k1; k += kThreads) p.out_row[k] = src[k];
    BYZ_STAMP(6);
}
#undef BYZ_STAMP

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
    const size_t slab_floats = static_cast<size_t>(ctx->num_cus) * kSlabFloats;
    BYZ_TRY(ctx->gram_partials.ensure((slab_floats + static_cast<size_t>(ctx->num_cus) * kMaxRows) * sizeof(float)));
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(kMaxRows) * sizeof(float)));
    return BYZ_OK;
}

// K1 + K2: dist (n x n fp32, pitch n) of the n_rows x n_cols matrix G and, with prefix_len >= 0, the Krum scores, the winner
// (winner_dev) and the copy of the winning row (out_row, optional)
static int small_round(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* dist,
                       int64_t prefix_len, int32_t* winner_dev, float* out_row, hipStream_t stream) {
    BYZ_REQUIRE(G && dist && n_rows >= 1 && n_rows <= kMaxRows && n_cols > 0 && ld >= n_cols,
                "small distances: bad shape %lld x %lld ld %lld", (long long)n_rows, (long long)n_cols, (long long)ld);
    const int n = static_cast<int>(n_rows);
    const int64_t n_slices = ceil_div(n_cols, kSlice);
    // one workgroup per CU at most; the grid is sized so that everybody gets the same number of slices (+- 1)
    // BYZ_KRUM_SMALL_GRID (experiments): a cap on K1's workgroups below 256 -- fewer, longer-lived workgroups, fewer slabs for K2
    int cap = env_int("BYZ_KRUM_SMALL_GRID", 256);
    cap = cap < 1 ? 1 : (cap > 256 ? 256 : cap);
    if (ctx->num_cus < cap) cap = ctx->num_cus;
    const int grid = static_cast<int>(ceil_div(n_slices, ceil_div(n_slices, cap)));
    const int64_t per = ceil_div(n_slices, grid);   // (per - 1) * grid < n_slices: only a workgroup's last slice can be ragged or missing
    // one full 128 x 128 slab per workgroup, then one compact diagonal per workgroup
    const size_t slab_floats = static_cast<size_t>(grid) * kSlabFloats;
    BYZ_TRY(ctx->gram_partials.ensure((slab_floats + static_cast<size_t>(grid) * kMaxRows) * sizeof(float)));
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(kMaxRows) * sizeof(float)));
    float* slabs = ctx->gram_partials.as<float>();
    float* diag_slabs = slabs + slab_floats;
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
        ctx->small_configured = true;
    }
    // BYZ_KRUM_SMALL_SKIP (timing experiments only, WRONG results): bit 0 no K1, bit 1 no K2 (scripts/c2_skip_probe.sh)
    const int skip = env_int("BYZ_KRUM_SMALL_SKIP", 0);
    if (!(skip & 1)) {
        KernelTimer t(ctx, BYZ_K_GRAM, stream);
        const int unrolled = env_int("BYZ_KRUM_SMALL_UNROLL", 1) != 0 && per <= 8 ? static_cast<int>(per) : 0;
#define BYZ_K1(T, P)                                                                               \
    small_gram_kernel<T, P><<<static_cast<unsigned>(T ? 1 : grid), kThreads, kGramLds, stream>>>(    \
        G, n, n_cols, ld, static_cast<int>(T ? 1 : n_slices), slabs, diag_slabs, ctx->scores.as<float>(), env_int("BYZ_KRUM_SMALL_DBG", 0))
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
    if (!(skip & 2)) {
        KernelTimer t(ctx, prefix_len >= 0 ? BYZ_K_ROW_SORT : BYZ_K_DISTANCES, stream);
        RowsArgs p;
        p.slabs = slabs;
        p.diag_slabs = diag_slabs;
        p.n_slabs = n_cols < kSlice ? 1 : grid;
        p.n = n;
        p.G = G;
        p.n_cols = n_cols;
        p.ld = ld;
        p.dist = dist;
        p.prefix_len = static_cast<int>(prefix_len);
        p.scores = ctx->scores.as<float>();
        p.winner = winner_dev;
        p.out_row = out_row;
        p.status = device_status_word(ctx);
        const bool stamps = env_int("BYZ_KRUM_SMALL_TIMING", 0) != 0;
        const int pg = 4 * static_cast<int>(ceil_div(p.n_slabs, 4 * kGroups));   // 4, 8, 12 or 16 slabs per thread group
#define BYZ_K2(TM)                                                                                 \
    switch (pg) {                                                                                 \
        case 4: small_rows_kernel<TM, 4><<<static_cast<unsigned>(n), kThreads, 0, stream>>>(p); break;   \
        case 8: small_rows_kernel<TM, 8><<<static_cast<unsigned>(n), kThreads, 0, stream>>>(p); break;   \
        case 12: small_rows_kernel<TM, 12><<<static_cast<unsigned>(n), kThreads, 0, stream>>>(p); break; \
        default: small_rows_kernel<TM, 16><<<static_cast<unsigned>(n), kThreads, 0, stream>>>(p); break; \
    }
        if (stamps) { BYZ_K2(true) } else { BYZ_K2(false) }
#undef BYZ_K2
        BYZ_TRY(check_launch("small_rows_kernel"));
        if (stamps) {
            static unsigned long long host[kMaxRows * kRowStamps];
            BYZ_HIP(hipStreamSynchronize(stream));
            BYZ_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rows_stamps), sizeof(host)));
            const char* names[7] = {"", "slab sums", "distances", "sort + score + ticket", "wait for all", "argmin", "row copy"};
            for (int k = 1; k <= 6; ++k) {
                if (k == 4 && prefix_len < 0) break;
                double sum = 0, mx = 0;
                for (int w = 0; w < n; ++w) {
                    const double dt = static_cast<double>(host[w * kRowStamps + k] - host[w * kRowStamps + k - 1]);
                    sum += dt;
                    if (dt > mx) mx = dt;
                }
                std::fprintf(stderr, "small_rows %-22s mean %8.0f  max %8.0f ticks\n", names[k], sum / n, mx);
            }
        }
    }
    return BYZ_OK;
}

int launch_small_distances(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* dist,
                           hipStream_t stream) {
    return small_round(ctx, G, n_rows, n_cols, ld, dist, -1, nullptr, nullptr, stream);
}

// The whole Krum round of defences.py:23-42 at N <= 128 in TWO launches: distances, scores, winner, the winning row's copy
int launch_small_krum(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* dist, int64_t prefix_len,
                      int32_t* winner_dev, float* out_row, hipStream_t stream) {
    BYZ_REQUIRE(winner_dev && prefix_len >= 0, "small krum: bad arguments");
    return small_round(ctx, G, n_rows, n_cols, ld, dist, prefix_len, winner_dev, out_row, stream);
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
