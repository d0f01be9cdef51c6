// Shared host-side plumbing of libbyzagg: the context, error reporting, workspace growth and the
// per-kernel event timing used by bench.py.  gfx950 only; no other target is considered.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/byzagg.h"

namespace byz {

void set_error(const char* fmt, ...);

#define BYZ_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ::byz::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                             __LINE__);                                                     \
            return BYZ_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

#define BYZ_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != BYZ_OK) return rc_; \
    } while (0)

#define BYZ_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ::byz::set_error(__VA_ARGS__); \
            return BYZ_E_INVALID;       \
        }                               \
    } while (0)

// A device buffer that only ever grows; growth happens between kernels, never inside the hot loop
// once byz_ctx_reserve has been called with the largest shape.
struct Buffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return BYZ_OK;
        if (ptr) BYZ_HIP(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
        BYZ_HIP(hipMalloc(&ptr, need));
        bytes = need;
        return BYZ_OK;
    }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(ptr); }
};

struct PinnedBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return BYZ_OK;
        if (ptr) BYZ_HIP(hipHostFree(ptr));
        ptr = nullptr;
        bytes = 0;
        BYZ_HIP(hipHostMalloc(&ptr, need, hipHostMallocDefault));
        bytes = need;
        return BYZ_OK;
    }
    void release() {
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};

struct TimingSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    int64_t launches = 0;
};

}  // namespace byz

struct byz_ctx {
    int device = 0;
    int num_cus = 256;
    // workspaces
    byz::Buffer gram_partials;   // split-K slabs of the Gram kernel
    byz::Buffer gram;            // n x n fp64 Gram
    byz::Buffer tile_order;      // (ti, tj) of every lower-triangle tile, in XCD-friendly order
    byz::Buffer dup_rep;         // representative row of every group of identical rows (+ one flag word)
    byz::Buffer gram_tickets;    // chunked Gram schedule: next chunk allowed to update a tile's slab
    byz::Buffer gram_planes;     // pre-split Gram: the bf16 planes of one super-chunk of columns, in MFMA fragment order
    byz::Buffer gram_chunk_sums; // pre-split Gram, f16x2: the fp32 level-1 sums of every (chunk, tile) of a launch (deferred slab update)
    byz::Buffer plane_unscale;   // pre-split Gram, f16x2: 2^-shift of every (chunk, row) of the super-chunk (fp64)
    byz::Buffer plane_order;     // pre-split Gram: (256-row block, 128-row block) of every workgroup tile
    byz::Buffer spec_owner;      // speculative Bulyan loop: the row every (workgroup, thread) slot owns
    byz::Buffer split_redo;      // pre-split Gram: (chunk, row block) pairs whose sampled scale did not hold (count first)
    std::vector<int32_t> plane_order_host;
    int64_t plane_order_T = -1;
    int64_t plane_order_share = 1 * 65536 + 0;
    byz::Buffer row_signature;   // dedup: 64-bit signature of every row
    byz::Buffer unique_rows;     // dedup: the unique rows, ascending
    byz::Buffer row_map;         // dedup: position of every row's representative among the unique rows (+ scratch)
    byz::Buffer gram_compact;    // dedup: Gram of the unique rows
    std::vector<int32_t> tile_order_host;
    int64_t tile_order_T = -1;
    int64_t tile_order_share = 1 * 65536 + 0;   // share_count * 65536 + share_index the list was built for
    byz::Buffer tile_owned;      // one byte per lower-triangle tile: 1 = in this launch's share
    int64_t row_map_rows = 0;    // dedup: row count the row_map currently describes (0: none)
    byz::Buffer gram_rep;        // first row with bitwise equal Gram entries (candidate identical row), per row
    byz::Buffer near_pairs;      // near-duplicate pairs (int2) whose distance is re-evaluated on the difference
    byz::Buffer near_sq;         // their squared distances (fp64)
    byz::Buffer near_rows;       // listed pairs per row (counts, then offsets): what makes the list's order canonical
    byz::Buffer near_partial;    // per (pair, column chunk) partial sums
    int64_t near_pair_capacity = 0;
    byz::Buffer dist;            // n x n fp32 distances (when the caller does not pass one)
    byz::Buffer sorted_idx;      // n x n uint16: column index at every ascending rank
    byz::Buffer sorted_val;      // n x n fp32: every row's distances in ascending order (the reference-arithmetic re-score)
    byz::Buffer rank_t;          // n x n uint16: rank_t[w][u] = rank of column w in row u
    byz::Buffer rank_rows;       // n x n uint16: the same table as the row sort writes it, rank_rows[u][w] (transposed into rank_t)
    byz::Buffer row_total;       // n fp64: sum of a row's finite distances
    byz::Buffer row_top;         // n fp64: sum of a row's largest `drop` finite distances; then n counts of non-finite ones
    byz::Buffer scores;          // n fp32 Krum scores
    // large_rows.hip: more than 16,384 rows
    byz::Buffer large_keys;      // 64-bit sort keys of one batch of rows / columns
    byz::Buffer large_idx;       // n x n uint32: column index at every ascending rank
    byz::Buffer large_rank;      // n x n uint32: rank of column w in row u, [u][w]
    byz::Buffer large_rank_t;    // n x n uint32: the same transposed, [w][u]: a pick reads the winner's row of it
    byz::Buffer large_dist_t;    // n x n fp32: the distance matrix transposed, [w][u] = d(u, w) (a caller's matrix need not be symmetric)
    byz::Buffer large_state;     // the Bulyan loop's per-row state
    byz::Buffer large_grid;      // its deciding kernel on many workgroups: every workgroup's best score, the list's counter
    bool redo_valid = false;     // the last trimmed mean went through the ring selection (redo_tiles[0] is its count)
    byz::Buffer redo_tiles;      // trimmed mean: tiles the ring selection handed to the general kernel (count first)
    byz::Buffer twin_class;      // 2n int32: twin class of every row (scratch, then final)
    byz::Buffer xchg;            // Bulyan grid loop: tagged 8-byte granules the workgroups exchange
    int64_t bulyan_rescored = 0; // rows the last Bulyan loop re-scored in the reference's fp32 arithmetic
    byz::Buffer selection;       // theta int32
    byz::Buffer small;           // misc device scalars (winner index, status words)
    bool small_configured = false;   // krum_small.hip: dynamic-LDS attributes set for this context's device
    // kernels whose hipFuncAttributeMaxDynamicSharedMemorySize has been raised for this context's device, and to what: the
    // attribute belongs to the (function, device) pair and setting it is a runtime call per launch otherwise (round 6)
    std::unordered_map<const void*, int> dynamic_lds;
    byz::Buffer assemble_table;  // byz_assemble_rows_dev: segment starts + every client's tensor pointers
    std::vector<int64_t> assemble_host;   // its host image: owned by the context, because an async copy out of pageable memory
    hipEvent_t assemble_copied = nullptr; // may still be reading it when the call returns; recorded behind that copy
    int64_t assemble_clients = 0, assemble_segments = 0, assemble_longest = 0, assemble_total = 0;   // what the device table holds
    byz::Buffer stage_in;        // device copy of a host matrix
    byz::Buffer stage_out;       // device result before download
    byz::PinnedBuffer pinned;    // host bounce buffer for small results
    // timing
    bool timing = false;
    byz::TimingSlot slots[BYZ_K_COUNT];
};

namespace byz {

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, context) and size, not once per launch
inline hipError_t allow_dynamic_lds(byz_ctx* ctx, const void* kernel, int bytes) {
    auto it = ctx->dynamic_lds.find(kernel);
    if (it != ctx->dynamic_lds.end() && it->second >= bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) ctx->dynamic_lds[kernel] = bytes;
    return e;
}

// ctx->small (256 bytes, allocated and zeroed with the context) holds the device-side scalars:
//   [0] Krum winner   [8] Bulyan loop status   [9] rows the Bulyan loop re-scored
//   [16] sticky device status (bit 0: a Gram chunk lost its ticket, bit 1: near-duplicate pair list overflowed,
//        bit 2: two rows with bitwise equal Gram entries turned out to differ, bit 3: the row workgroups of the small-N
//        path did not all report their scores in time, bit 4: a wave of the register-resident column statistics never got
//        its turn)
//   [17] number of near-duplicate pairs listed by the last distance kernel
//   [18] non-zero: the register-resident column statistics of the CURRENT call gave up a turn; the two-pass kernel queued
//        behind them recomputes the call's columns (zeroed before every such launch; bit 4 above is no longer set)
inline int32_t* device_status_word(byz_ctx* ctx) { return ctx->small.as<int32_t>() + 16; }
inline int32_t* attack_redo_word(byz_ctx* ctx) { return ctx->small.as<int32_t>() + 18; }
inline int32_t* near_pair_count_word(byz_ctx* ctx) { return ctx->small.as<int32_t>() + 17; }

// Brackets one kernel launch with events when timing is on (bench.py's roofline leg).
struct KernelTimer {
    byz_ctx* ctx;
    int kernel;
    hipStream_t stream;
    hipEvent_t start = nullptr, stop = nullptr;
    KernelTimer(byz_ctx* c, int k, hipStream_t s) : ctx(c), kernel(k), stream(s) {
        if (ctx->timing) {
            (void)hipEventCreate(&start);
            (void)hipEventCreate(&stop);
            (void)hipEventRecord(start, stream);
        }
    }
    ~KernelTimer() {
        if (ctx->timing && start) {
            (void)hipEventRecord(stop, stream);
            ctx->slots[kernel].pending.emplace_back(start, stop);
        }
    }
};

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return BYZ_E_HIP;
    }
    return BYZ_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t next_pow2(int64_t v) {
    int64_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

// ---- kernel launchers (one per .hip file) -------------------------------------------------------
int launch_column_mean(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                       float* out, hipStream_t stream);
int launch_column_drift(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                        float num_std, float* drift, float* mean, float* stdev, hipStream_t stream);
int launch_column_chain(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const float* carry,
                        const float* mean, float* out, hipStream_t stream);
int launch_column_finish(byz_ctx* ctx, const float* sum, const float* sumsq, int64_t total_rows, float num_std, int64_t n_cols,
                         float* mean, float* stdev, float* drift, hipStream_t stream);
int launch_broadcast_rows(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                          const float* vec, hipStream_t stream);
int launch_drift_axpy(byz_ctx* ctx, float* mean, const float* stdev, int64_t n, float num_std,
                      hipStream_t stream);
int launch_server_update(byz_ctx* ctx, float* w, float* v, const float* agg, int64_t n, float momentum,
                         float lr, hipStream_t stream);
int launch_copy_row(byz_ctx* ctx, const float* G, int64_t ld, int64_t n_rows, int64_t n_cols,
                    const int32_t* index_dev, float* out, hipStream_t stream);

// round_edges.hip: the steps either side of the path
constexpr int kMaxSegments = 32;   // tensors one assemble launch can place (more take further launches)
int launch_backdoor_initial(byz_ctx* ctx, const float* params, const float* mean, int64_t n, float lr, float* out,
                            hipStream_t stream);
int launch_backdoor_clip(byz_ctx* ctx, const float* mean, const float* stdev, const float* params, const float* mal,
                         int64_t n, float lr, float z, float* out, hipStream_t stream);
int launch_assemble_row(byz_ctx* ctx, float* row, int64_t n_cols, int64_t n_segments, const float* const* segments,
                        const int64_t* lengths, hipStream_t stream);
int launch_assemble_columns(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t n_segments,
                            const float* const* segments, const int64_t* lengths, hipStream_t stream);
int launch_assemble_rows_again(byz_ctx* ctx, float* G, int64_t n_cols, int64_t ld, int64_t n_clients, int64_t n_segments,
                               hipStream_t stream);
int launch_assemble_rows(byz_ctx* ctx, float* G, int64_t n_cols, int64_t ld, int64_t n_clients, int64_t n_segments,
                         const float* const* segments, const int64_t* lengths, hipStream_t stream);

int launch_gram(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, double* gram,
                hipStream_t stream);
// gram_planes.hip: the long-K Gram on operands split once into bf16 planes
bool gram_planes_enabled();
int launch_gram_planes(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                       double* slabs, int share_count, int share_index, uint8_t* owned_host, bool f16, hipStream_t stream);
int launch_gram_share(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                      int share_count, int share_index, double* gram, hipStream_t stream, bool accumulate = false);
// dedup.hip: identical rows found before the Gram
int find_unique_rows(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, hipStream_t stream,
                     int64_t* n_unique_host);
int launch_gram_expand(byz_ctx* ctx, const double* compact, int64_t n_unique, int64_t n_rows, double* gram,
                       hipStream_t stream);
int launch_distances_from_gram(byz_ctx* ctx, const double* gram, int64_t n, float* dist, hipStream_t stream,
                               const float* G, int64_t n_cols, int64_t ld);
int launch_near_pair_sqdist(byz_ctx* ctx, const float* G, int64_t n_cols, int64_t ld, const int32_t* row_index,
                            double* sq_dev, hipStream_t stream);
int launch_near_pair_apply(byz_ctx* ctx, const double* sq_dev, int64_t n, float* dist, hipStream_t stream);

int launch_row_sort(byz_ctx* ctx, const float* dist, int64_t n, int64_t prefix_len, int64_t drop_count,
                    bool want_tables, hipStream_t stream);
int launch_krum_argmin(byz_ctx* ctx, int64_t n, int32_t* winner_dev, hipStream_t stream);
int launch_bulyan_loop(byz_ctx* ctx, const float* dist, int64_t n, int64_t theta, int64_t drop_count,
                       int64_t users_count, int64_t corrupted, int32_t* selection_dev, int32_t* status_dev,
                       hipStream_t stream);

// large_rows.hip: beyond the LDS-resident kernels' 16,384 rows (BYZ_SELECT_LARGE=1 / BYZ_TM_LARGE=1 take these paths at any size)
constexpr int64_t kLargeMaxRows = int64_t{1} << 20;
bool select_large_applies(int64_t n);
int segment_sort_u64(byz_ctx* ctx, unsigned long long* keys, int64_t n_segments, int64_t n_pad, hipStream_t stream);
int segment_sort_u32(byz_ctx* ctx, uint32_t* keys, int64_t n_segments, int64_t n_pad, hipStream_t stream);
size_t large_key_scratch_bytes();
int launch_row_sort_large(byz_ctx* ctx, const float* dist, int64_t n, int64_t prefix_len, int64_t drop_count, bool want_tables,
                          hipStream_t stream);
int launch_bulyan_loop_large(byz_ctx* ctx, const float* dist, int64_t n, int64_t theta, int64_t drop_count, int64_t users_count,
                             int64_t corrupted, const int32_t* twin_class, int32_t* selection_dev, int32_t* status_dev,
                             hipStream_t stream);

int launch_trimmed_mean(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                        const int32_t* row_index, int64_t keep, float* out, hipStream_t stream);
int64_t trimmed_mean_max_rows();
bool trimmed_mean_large_applies(int64_t n_rows);
// tall_select.hip: more rows than the register kernels hold (5,632): order statistics by radix select, the column streamed from HBM
bool trimmed_mean_tall_applies(int64_t n_rows);
int launch_trimmed_mean_tall(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                             int64_t keep, float* out, hipStream_t stream);
int launch_trimmed_mean_large(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                              int64_t keep, float* out, hipStream_t stream);
// window_lean.hip: the row-split ring selection, first stage of the trimmed mean (round 3)
int launch_window_lean(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const int32_t* row_index,
                       int64_t keep, float* out, int32_t* redo, hipStream_t stream);
int64_t select_max_rows();

int launch_lane_selftest(byz_ctx* ctx, int32_t* out, int32_t* n_patterns, hipStream_t stream);

// krum_small.hip: the whole of Krum for N <= 128 in five launches
bool krum_small_applies(int64_t n_rows, int64_t n_cols);
int reserve_small_workspaces(byz_ctx* ctx);
int launch_small_distances(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* dist,
                           hipStream_t stream);
int launch_small_krum(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* dist, int64_t prefix_len,
                      int32_t* winner_dev, float* out_row, hipStream_t stream);
int launch_small_select(byz_ctx* ctx, const float* dist, int64_t n_rows, int64_t prefix_len, const float* G, int64_t n_cols,
                        int64_t ld, int32_t* winner_dev, float* out_row, hipStream_t stream);
}  // namespace byz
