// Identical client rows, found before the Gram (reference malicious.py:26-27: every malicious client submits the SAME
// vector, so under the attack m = f of the N rows are bitwise equal; BASELINE configs[4]: 2,400 of 10,000).
//
// The Gram costs N^2 D / 2 multiply-adds and nothing else on the path comes close, so the rows are deduplicated first:
//   1. signature   64-bit hash of 8 KiB sampled from every row (eight 1 KiB segments spread over the columns);
//   2. candidates  rep[i] = first j <= i with the same signature (the signatures in front pass through LDS a tile at a time);
//   3. verify      every candidate row is compared with its representative bit for bit over ALL columns; a row that
//                  differs anywhere is its own representative again (signatures only nominate, they never decide);
//   4. compact     the unique rows in ascending order, and for every row the position of its representative there.
// The Gram then runs over the U unique rows (row indirection in gram_tile_kernel) and is expanded to N x N:
// identical rows get bitwise identical Gram rows by construction, everything downstream is unchanged.
// Cost without duplicates: 8 KiB per row read, one small scan, one 4-byte read-back (~0.1 ms at N = 10,000).
// With configs[4]'s duplicates: the verification reads the 2,400 duplicate rows once (+1.5% of the matrix) and the
// Gram shrinks from 79^2/2 to 60^2/2 tiles (-42%).
#include "common.hpp"

namespace byz {
namespace {

constexpr int kSegments = 8;          // sampled segments per row
constexpr int kSegmentFloats = 256;   // 1 KiB each: one coalesced load per thread

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// one workgroup (256 threads) per row; the per-element terms are summed, so the reduction order does not matter
__global__ __launch_bounds__(256) void row_signature_kernel(const float* __restrict__ G, int64_t n_cols, int64_t ld,
                                                            uint64_t* __restrict__ signature) {
    __shared__ uint64_t part[4];
    const int64_t row = blockIdx.x;
    const uint32_t* __restrict__ bits = reinterpret_cast<const uint32_t*>(G + row * ld);
    const int64_t span = n_cols > kSegmentFloats ? n_cols - kSegmentFloats : 0;
    uint64_t h = 0;
#pragma unroll
    for (int s = 0; s < kSegments; ++s) {
        const int64_t col = (span * s) / (kSegments - 1) + threadIdx.x;
        if (col < n_cols && (s == 0 || span > 0))
            h += mix64((static_cast<uint64_t>(bits[col]) << 24) ^ static_cast<uint64_t>(col) * 0x9e3779b97f4a7c15ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) signature[row] = (part[0] + part[1]) + (part[2] + part[3]);
}

// rep[i] = first row with row i's signature (itself when there is none before it); rows that found one are listed
__global__ __launch_bounds__(256) void candidate_kernel(const uint64_t* __restrict__ signature, int n,
                                                        int32_t* __restrict__ rep, int32_t* __restrict__ mismatch,
                                                        int32_t* __restrict__ candidates, int32_t* __restrict__ n_candidates) {
    // (Round 6: the signatures of the rows in front pass through LDS a tile at a time and every thread looks at all of a tile.  The
    // first form walked signature[0 .. i) out of global memory with a break at the first match -- a chain of dependent loads as
    // long as the row index: 0.37 ms at N = 4000, a millisecond at 10,000, for a kernel that compares n^2 / 2 words.)
    __shared__ uint64_t tile[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const uint64_t mine = i < n ? signature[i] : 0ull;
    int r = i;
    const int last = blockIdx.x * 256 + 255;          // the workgroup's largest row index: tiles beyond it hold no row in front
    for (int j0 = 0; j0 <= last && j0 < n; j0 += 256) {
        tile[threadIdx.x] = j0 + threadIdx.x < n ? signature[j0 + threadIdx.x] : 0ull;
        __syncthreads();
        const int m = n - j0 < 256 ? n - j0 : 256;
#pragma unroll 8
        for (int j = 0; j < m; ++j) {
            const int cand = j0 + j;
            if (tile[j] == mine && cand < r) r = cand;      // the smallest index with this signature (r starts at i: only rows in front)
        }
        __syncthreads();
    }
    if (i >= n) return;
    rep[i] = r;
    mismatch[i] = 0;
    if (r != i) candidates[atomicAdd(n_candidates, 1)] = i;   // the order of the list does not matter
}

// Bitwise comparison of every candidate row with its representative.  A fixed grid walks the candidate list
// (blockIdx.y strides the candidates, blockIdx.x the columns), so a matrix without duplicates costs 2,048 idle workgroups.
template <bool VEC>
__global__ __launch_bounds__(256) void verify_kernel(const float* __restrict__ G, int64_t n_cols, int64_t ld,
                                                     const int32_t* __restrict__ rep,
                                                     const int32_t* __restrict__ candidates,
                                                     const int32_t* __restrict__ n_candidates,
                                                     int32_t* __restrict__ mismatch) {
    const int count = *n_candidates;
    const int64_t first = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int c = blockIdx.y; c < count; c += gridDim.y) {
        const int64_t i = candidates[c];
        const int64_t p = rep[i];
        bool differs = false;
        const uint32_t* __restrict__ as = reinterpret_cast<const uint32_t*>(G + i * ld);
        const uint32_t* __restrict__ bs = reinterpret_cast<const uint32_t*>(G + p * ld);
        if constexpr (VEC) {
            const uint4* __restrict__ a = reinterpret_cast<const uint4*>(as);
            const uint4* __restrict__ b = reinterpret_cast<const uint4*>(bs);
            const int64_t quads = n_cols / 4;
            for (int64_t q = first; q < quads; q += stride) {
                const uint4 x = a[q], y = b[q];
                differs = differs || x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
            }
            const int64_t k = quads * 4 + first;
            if (k < n_cols) differs = differs || as[k] != bs[k];   // fewer than four ragged columns
        } else {
            for (int64_t k = first; k < n_cols; k += stride) differs = differs || as[k] != bs[k];
        }
        if (differs) mismatch[i] = 1;   // same value from every writer
    }
}

// One workgroup: unique rows in ascending order, and map[i] = position of row i's representative among them.
// Rows whose verification failed are their own representatives.
__global__ __launch_bounds__(1024) void compact_kernel(int n, int32_t* __restrict__ rep, const int32_t* __restrict__ mismatch,
                                                       int32_t* __restrict__ unique_rows, int32_t* __restrict__ map,
                                                       int32_t* __restrict__ n_unique) {
    __shared__ int wave_total[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        bool unique = false;
        if (i < n) {
            if (mismatch[i]) rep[i] = i;
            unique = rep[i] == i;
        }
        const unsigned long long ballot = __ballot(unique);
        const int before = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_total[wave] = __popcll(ballot);
        __syncthreads();
        int offset = carry;
        for (int w = 0; w < wave; ++w) offset += wave_total[w];
        if (unique) {
            unique_rows[offset + before] = i;
            map[i] = offset + before;
        }
        __syncthreads();
        if (tid == 0) {
            int total = 0;
            for (int w = 0; w < 16; ++w) total += wave_total[w];
            carry += total;
        }
        __syncthreads();
    }
    // a representative always precedes its duplicates, so its position is already written
    for (int i = tid; i < n; i += 1024)
        if (rep[i] != i) map[i] = map[rep[i]];
    if (tid == 0) *n_unique = carry;
}

__global__ __launch_bounds__(256) void gram_expand_kernel(const double* __restrict__ compact, int64_t n_unique,
                                                          const int32_t* __restrict__ map, int64_t n,
                                                          double* __restrict__ gram) {
    const int64_t j = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (j >= n) return;
    const int64_t mj = map[j];
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y) gram[i * n + j] = compact[static_cast<int64_t>(map[i]) * n_unique + mj];
}

}  // namespace

// Fills ctx->unique_rows (n_unique int32, ascending) and ctx->row_map (n int32); *n_unique_host = U.  Synchronises
// the stream once (the caller sizes the Gram launch with U).
int find_unique_rows(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, hipStream_t stream,
                     int64_t* n_unique_host) {
    BYZ_REQUIRE(n_rows < (int64_t{1} << 24), "dedup: too many rows");
    const int n = static_cast<int>(n_rows);
    BYZ_TRY(ctx->row_signature.ensure(static_cast<size_t>(n) * sizeof(uint64_t)));
    BYZ_TRY(ctx->unique_rows.ensure(static_cast<size_t>(n) * sizeof(int32_t)));
    // row_map: [map (n) | rep (n) | mismatch (n) | candidates (n) | n_unique, n_candidates]
    BYZ_TRY(ctx->row_map.ensure(static_cast<size_t>(4 * n + 2) * sizeof(int32_t)));
    int32_t* map = ctx->row_map.as<int32_t>();
    int32_t* rep = map + n;
    int32_t* mismatch = rep + n;
    int32_t* candidates = mismatch + n;
    int32_t* count = candidates + n;
    int32_t* n_candidates = count + 1;
    uint64_t* sig = ctx->row_signature.as<uint64_t>();
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    BYZ_HIP(hipMemsetAsync(n_candidates, 0, sizeof(int32_t), stream));
    row_signature_kernel<<<static_cast<unsigned>(n), 256, 0, stream>>>(G, n_cols, ld, sig);
    BYZ_TRY(check_launch("row_signature_kernel"));
    candidate_kernel<<<static_cast<unsigned>(ceil_div(n, 256)), 256, 0, stream>>>(sig, n, rep, mismatch, candidates, n_candidates);
    BYZ_TRY(check_launch("candidate_kernel"));
    {
        int64_t blocks = ceil_div(n_cols, static_cast<int64_t>(256) * 4 * 8);   // >= 8 vector loads per thread and row
        if (blocks < 1) blocks = 1;
        if (blocks > 64) blocks = 64;
        const dim3 grid(static_cast<unsigned>(blocks), 32);
        const bool vec = (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0);
        if (vec) verify_kernel<true><<<grid, 256, 0, stream>>>(G, n_cols, ld, rep, candidates, n_candidates, mismatch);
        else verify_kernel<false><<<grid, 256, 0, stream>>>(G, n_cols, ld, rep, candidates, n_candidates, mismatch);
        BYZ_TRY(check_launch("verify_kernel"));
    }
    compact_kernel<<<1, 1024, 0, stream>>>(n, rep, mismatch, ctx->unique_rows.as<int32_t>(), map, count);
    BYZ_TRY(check_launch("compact_kernel"));
    BYZ_TRY(ctx->pinned.ensure(sizeof(int32_t)));
    BYZ_HIP(hipMemcpyAsync(ctx->pinned.ptr, count, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    BYZ_HIP(hipStreamSynchronize(stream));
    *n_unique_host = *static_cast<const int32_t*>(ctx->pinned.ptr);
    return BYZ_OK;
}

int launch_gram_expand(byz_ctx* ctx, const double* compact, int64_t n_unique, int64_t n_rows, double* gram,
                       hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_GRAM_REDUCE, stream);
    const dim3 grid(static_cast<unsigned>(ceil_div(n_rows, 256)), static_cast<unsigned>(n_rows < 32768 ? n_rows : 32768));
    gram_expand_kernel<<<grid, 256, 0, stream>>>(compact, n_unique, ctx->row_map.as<int32_t>(), n_rows, gram);
    return check_launch("gram_expand_kernel");
}

}  // namespace byz
