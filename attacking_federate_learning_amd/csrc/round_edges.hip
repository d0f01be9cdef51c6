// The steps either side of the aggregation path, each a single HBM-bound pass.
//
//   backdoor hook     reference backdoor.py:52-65   BackdoorAttack._attack_grads without its training loop:
//                                                   the start parameters of the malicious network, and the
//                                                   clip of the resulting gradient to mean +- z * std
//   gradient assembly reference user.py:92,         a client's per-parameter gradient tensors written straight into
//                     server.py:81-83               its row of the device-resident N x D matrix
//
// Arithmetic of the backdoor hook is numpy's, operation by operation in fp32 (no contraction into FMAs, IEEE
// division), so the results are bit-identical to the reference's on the same inputs:
//     t        = lr * mean
//     initial  = params - t                               backdoor.py:54
//     new      = mal_net + t                              backdoor.py:59
//     grads    = (initial - new) / lr                     backdoor.py:60
//     out      = clip(grads, mean - z * std, mean + z * std)   backdoor.py:62-63
// np.clip is minimum(maximum(x, lo), hi) with NaN propagating from either operand.
//
// Bytes per element: initial 12 (2 reads, 1 write); clip 20 (4 reads, 1 write); assembly 8 (1 read, 1 write).
#include "common.hpp"

#include <vector>

// hipcc contracts a * b - c into an FMA by default (-ffp-contract=fast, and the __fmul_rn / __fsub_rn wrappers do not
// stop it: they are inline functions carrying the same flag); numpy rounds the product first.  This file is built
// with -ffp-contract=off (build_native.py) and says so here as well.
#pragma clang fp contract(off)

namespace byz {
namespace {

constexpr int kThreads = 256;
constexpr int kVec = 4;   // elements per thread and pass in the vector kernels

__device__ __forceinline__ float np_maximum(float a, float b) { return (a != a || a > b) ? a : b; }   // numpy: NaN wins
__device__ __forceinline__ float np_minimum(float a, float b) { return (a != a || a < b) ? a : b; }

__device__ __forceinline__ float clip_one(float mean, float stdev, float params, float mal, float lr, float z) {
    const float t = lr * mean;
    const float initial = params - t;
    const float renewed = mal + t;
    const float grads = (initial - renewed) / lr;   // IEEE division (hipcc's default for fp32 '/')
    const float band = z * stdev;
    return np_minimum(np_maximum(grads, mean - band), mean + band);
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void backdoor_initial_kernel(const float* __restrict__ params,
                                                                    const float* __restrict__ mean, int64_t n,
                                                                    float lr, float* __restrict__ out) {
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * (VEC ? kVec : 1);
    if constexpr (VEC) {
        if (i + kVec <= n) {
            const float4 p = *reinterpret_cast<const float4*>(params + i);
            const float4 m = *reinterpret_cast<const float4*>(mean + i);
            float4 o;
            o.x = p.x - lr * m.x;
            o.y = p.y - lr * m.y;
            o.z = p.z - lr * m.z;
            o.w = p.w - lr * m.w;
            *reinterpret_cast<float4*>(out + i) = o;
            return;
        }
        for (int64_t k = i; k < n; ++k) out[k] = params[k] - lr * mean[k];
    } else {
        if (i < n) out[i] = params[i] - lr * mean[i];
    }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void backdoor_clip_kernel(const float* __restrict__ mean,
                                                                 const float* __restrict__ stdev,
                                                                 const float* __restrict__ params,
                                                                 const float* __restrict__ mal, int64_t n, float lr,
                                                                 float z, float* __restrict__ out) {
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x) * (VEC ? kVec : 1);
    if constexpr (VEC) {
        if (i + kVec <= n) {
            const float4 m = *reinterpret_cast<const float4*>(mean + i);
            const float4 s = *reinterpret_cast<const float4*>(stdev + i);
            const float4 p = *reinterpret_cast<const float4*>(params + i);
            const float4 q = *reinterpret_cast<const float4*>(mal + i);
            float4 o;
            o.x = clip_one(m.x, s.x, p.x, q.x, lr, z);
            o.y = clip_one(m.y, s.y, p.y, q.y, lr, z);
            o.z = clip_one(m.z, s.z, p.z, q.z, lr, z);
            o.w = clip_one(m.w, s.w, p.w, q.w, lr, z);
            *reinterpret_cast<float4*>(out + i) = o;
            return;
        }
        for (int64_t k = i; k < n; ++k) out[k] = clip_one(mean[k], stdev[k], params[k], mal[k], lr, z);
    } else {
        if (i < n) out[i] = clip_one(mean[i], stdev[i], params[i], mal[i], lr, z);
    }
}

// One launch copies up to kMaxSegments tensors into consecutive ranges of one row.  The table travels in the kernel
// arguments (no host-to-device copy, nothing to keep alive); blockIdx.y is the segment.
struct SegmentTable {
    const float* src[kMaxSegments];
    int64_t start[kMaxSegments + 1];   // first column of every segment, and the end of the last one
};

// dst[0 .. len) = src[0 .. len), cooperatively: thread `first` of `stride`.  16-byte moves when source and destination
// agree on alignment; the ragged head and tail go one by one.
__device__ __forceinline__ void copy_span(const float* __restrict__ src, float* __restrict__ dst, int64_t len,
                                          int64_t first, int64_t stride) {
    const uintptr_t sa = reinterpret_cast<uintptr_t>(src), da = reinterpret_cast<uintptr_t>(dst);
    if (((sa ^ da) & 15u) == 0 && len >= 8) {
        const int64_t head = ((16 - (sa & 15u)) & 15u) / 4;
        const int64_t quads = (len - head) / 4;
        const int64_t tail = head + quads * 4;
        for (int64_t q = first; q < quads; q += stride)
            *reinterpret_cast<float4*>(dst + head + 4 * q) = *reinterpret_cast<const float4*>(src + head + 4 * q);
        if (first < head) dst[first] = src[first];
        if (tail + first < len) dst[tail + first] = src[tail + first];   // fewer than 4 elements
    } else {
        for (int64_t k = first; k < len; k += stride) dst[k] = src[k];
    }
}

// One client: segment s is that client's gradient of parameter s.  blockIdx.y = segment.
__global__ __launch_bounds__(kThreads) void assemble_row_kernel(SegmentTable table, float* __restrict__ row) {
    const int seg = blockIdx.y;
    const int64_t begin = table.start[seg], len = table.start[seg + 1] - begin;
    copy_span(table.src[seg], row + begin, len, static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x,
              static_cast<int64_t>(gridDim.x) * kThreads);
}

// All clients at once: segment s is the (n_rows x len_s) row-major gradient of parameter s for every client (what a
// batched client step produces).  blockIdx.y = client row, blockIdx.z = segment.
__global__ __launch_bounds__(kThreads) void assemble_columns_kernel(SegmentTable table, float* __restrict__ G,
                                                                    int64_t ld) {
    const int seg = blockIdx.z;
    const int64_t r = blockIdx.y;
    const int64_t begin = table.start[seg], len = table.start[seg + 1] - begin;
    copy_span(table.src[seg] + r * len, G + r * ld + begin, len,
              static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x, static_cast<int64_t>(gridDim.x) * kThreads);
}

// Many clients, each with its OWN tensors (the reference's per-client loop, server.py:81-83, in one launch): the table lives
// in device memory -- starts[0 .. n_segments], then n_clients x n_segments pointers, client-major.  blockIdx.y = client,
// blockIdx.z = segment.
__global__ __launch_bounds__(kThreads) void assemble_rows_kernel(const int64_t* __restrict__ starts,
                                                                 const float* const* __restrict__ src, int n_segments,
                                                                 float* __restrict__ G, int64_t ld) {
    const int seg = blockIdx.z;
    const int64_t r = blockIdx.y;
    const int64_t begin = starts[seg], len = starts[seg + 1] - begin;
    copy_span(src[r * n_segments + seg], G + r * ld + begin, len, static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x,
              static_cast<int64_t>(gridDim.x) * kThreads);
}

bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace

int launch_backdoor_initial(byz_ctx* ctx, const float* params, const float* mean, int64_t n, float lr, float* out,
                            hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    if (aligned16(params) && aligned16(mean) && aligned16(out)) {
        backdoor_initial_kernel<true><<<static_cast<unsigned>(ceil_div(n, static_cast<int64_t>(kThreads) * kVec)), kThreads, 0, stream>>>(params, mean, n, lr, out);
    } else {
        backdoor_initial_kernel<false><<<static_cast<unsigned>(ceil_div(n, kThreads)), kThreads, 0, stream>>>(params, mean, n, lr, out);
    }
    return check_launch("backdoor_initial_kernel");
}

int launch_backdoor_clip(byz_ctx* ctx, const float* mean, const float* stdev, const float* params, const float* mal,
                         int64_t n, float lr, float z, float* out, hipStream_t stream) {
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    if (aligned16(mean) && aligned16(stdev) && aligned16(params) && aligned16(mal) && aligned16(out)) {
        backdoor_clip_kernel<true><<<static_cast<unsigned>(ceil_div(n, static_cast<int64_t>(kThreads) * kVec)), kThreads, 0, stream>>>(mean, stdev, params, mal, n, lr, z, out);
    } else {
        backdoor_clip_kernel<false><<<static_cast<unsigned>(ceil_div(n, kThreads)), kThreads, 0, stream>>>(mean, stdev, params, mal, n, lr, z, out);
    }
    return check_launch("backdoor_clip_kernel");
}

// rows == 0: one client (`dst` is its row); rows > 0: every client (`dst` is G, segments are rows x len blocks)
static int assemble(byz_ctx* ctx, float* dst, int64_t rows, int64_t ld, int64_t n_cols, int64_t n_segments,
                    const float* const* segments, const int64_t* lengths, hipStream_t stream) {
    int64_t total = 0;
    for (int64_t s = 0; s < n_segments; ++s) {
        BYZ_REQUIRE(segments[s] && lengths[s] >= 0, "assemble: bad segment %lld", (long long)s);
        total += lengths[s];
    }
    BYZ_REQUIRE(total == n_cols, "assemble: segments hold %lld values per client, a row %lld", (long long)total,
                (long long)n_cols);
    BYZ_REQUIRE(rows <= 65535, "assemble: at most 65535 clients per call, got %lld", (long long)rows);
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    int64_t column = 0;
    for (int64_t s0 = 0; s0 < n_segments; s0 += kMaxSegments) {
        SegmentTable table;
        const int count = static_cast<int>(n_segments - s0 < kMaxSegments ? n_segments - s0 : kMaxSegments);
        int64_t longest = 1;
        for (int k = 0; k < kMaxSegments; ++k) {
            table.src[k] = k < count ? segments[s0 + k] : nullptr;
            table.start[k] = column;
            if (k < count) {
                column += lengths[s0 + k];
                if (lengths[s0 + k] > longest) longest = lengths[s0 + k];
            }
        }
        table.start[kMaxSegments] = column;
        // one pass of 16-byte moves covers 1024 values per workgroup; cap the grid, the kernel strides
        int64_t blocks = ceil_div(longest, static_cast<int64_t>(kThreads) * 4);
        const int64_t cap = static_cast<int64_t>(ctx->num_cus) * 8;
        if (blocks > cap) blocks = cap;
        if (rows > 0) {
            assemble_columns_kernel<<<dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(rows), static_cast<unsigned>(count)), kThreads, 0, stream>>>(table, dst, ld);
        } else {
            assemble_row_kernel<<<dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(count)), kThreads, 0, stream>>>(table, dst);
        }
        BYZ_TRY(check_launch("assemble kernel"));
    }
    return BYZ_OK;
}

int launch_assemble_rows(byz_ctx* ctx, float* G, int64_t n_cols, int64_t ld, int64_t n_clients, int64_t n_segments,
                         const float* const* segments, const int64_t* lengths, hipStream_t stream) {
    BYZ_REQUIRE(n_clients <= 65535 && n_segments <= 65535, "assemble_rows: at most 65535 clients and tensors per call");
    // The table's host image belongs to the context: hipMemcpyAsync out of pageable memory may pin the pages and copy later
    // (large tables: thousands of clients), so the source must outlive this call; the event behind the copy says when the
    // image may be rewritten.
    if (ctx->assemble_copied == nullptr) BYZ_HIP(hipEventCreateWithFlags(&ctx->assemble_copied, hipEventDisableTiming));
    else BYZ_HIP(hipEventSynchronize(ctx->assemble_copied));
    std::vector<int64_t>& table = ctx->assemble_host;
    table.assign(static_cast<size_t>(n_segments + 1 + n_clients * n_segments), 0);
    int64_t total = 0, longest = 1;
    for (int64_t s = 0; s < n_segments; ++s) {
        BYZ_REQUIRE(lengths[s] >= 0, "assemble_rows: bad length of tensor %lld", (long long)s);
        table[static_cast<size_t>(s)] = total;
        total += lengths[s];
        if (lengths[s] > longest) longest = lengths[s];
    }
    table[static_cast<size_t>(n_segments)] = total;
    BYZ_REQUIRE(total == n_cols, "assemble_rows: a client's tensors hold %lld values, a row %lld", (long long)total,
                (long long)n_cols);
    for (int64_t k = 0; k < n_clients * n_segments; ++k) {
        BYZ_REQUIRE(segments[k] != nullptr || lengths[k % n_segments] == 0, "assemble_rows: null tensor (client %lld, tensor %lld)",
                    (long long)(k / n_segments), (long long)(k % n_segments));
        table[static_cast<size_t>(n_segments + 1 + k)] = static_cast<int64_t>(reinterpret_cast<uintptr_t>(segments[k]));
    }
    BYZ_TRY(ctx->assemble_table.ensure(table.size() * sizeof(int64_t)));
    BYZ_HIP(hipMemcpyAsync(ctx->assemble_table.ptr, table.data(), table.size() * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    BYZ_HIP(hipEventRecord(ctx->assemble_copied, stream));
    ctx->assemble_clients = n_clients;
    ctx->assemble_segments = n_segments;
    ctx->assemble_longest = longest;
    ctx->assemble_total = total;
    return launch_assemble_rows_again(ctx, G, n_cols, ld, n_clients, n_segments, stream);
}

// The launch alone, from the table the last launch_assemble_rows left on the device.
int launch_assemble_rows_again(byz_ctx* ctx, float* G, int64_t n_cols, int64_t ld, int64_t n_clients, int64_t n_segments,
                               hipStream_t stream) {
    BYZ_REQUIRE(ctx->assemble_clients > 0 && ctx->assemble_clients == n_clients && ctx->assemble_segments == n_segments &&
                    ctx->assemble_total == n_cols,
                "assemble_rows_again: the device table holds %lld clients x %lld tensors of %lld values, the call says %lld x %lld of %lld",
                (long long)ctx->assemble_clients, (long long)ctx->assemble_segments, (long long)ctx->assemble_total,
                (long long)n_clients, (long long)n_segments, (long long)n_cols);
    KernelTimer t(ctx, BYZ_K_MISC, stream);
    int64_t blocks = ceil_div(ctx->assemble_longest, static_cast<int64_t>(kThreads) * 4);
    const int64_t cap = 64;     // n_clients x n_segments workgroup columns already fill the chip; the kernel strides
    if (blocks > cap) blocks = cap;
    const int64_t* starts = ctx->assemble_table.as<int64_t>();
    assemble_rows_kernel<<<dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(n_clients), static_cast<unsigned>(n_segments)),
                           kThreads, 0, stream>>>(starts, reinterpret_cast<const float* const*>(starts + n_segments + 1),
                                                  static_cast<int>(n_segments), G, ld);
    return check_launch("assemble_rows_kernel");
}

int launch_assemble_row(byz_ctx* ctx, float* row, int64_t n_cols, int64_t n_segments, const float* const* segments,
                        const int64_t* lengths, hipStream_t stream) {
    return assemble(ctx, row, 0, 0, n_cols, n_segments, segments, lengths, stream);
}

int launch_assemble_columns(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t n_segments,
                            const float* const* segments, const int64_t* lengths, hipStream_t stream) {
    return assemble(ctx, G, n_rows, ld, n_cols, n_segments, segments, lengths, stream);
}

}  // namespace byz
