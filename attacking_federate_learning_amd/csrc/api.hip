// extern "C" surface of libbyzagg (include/byzagg.h): argument checks mirroring the reference's asserts,
// workspace management, and the composition of the kernels into the reference's functions.
#include "common.hpp"

#include <cstring>

namespace byz {

static thread_local char g_error[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

namespace {

// number of items `some_list[:stop]` keeps for a list of `length` items (Python slice semantics, which
// the reference relies on in defences.py:34 and :50)
int64_t python_prefix_len(int64_t length, int64_t stop) {
    if (stop >= 0) return stop < length ? stop : length;
    return length + stop > 0 ? length + stop : 0;
}

int enter(byz_ctx* ctx) {
    if (!ctx) {
        set_error("null context");
        return BYZ_E_INVALID;
    }
    BYZ_HIP(hipSetDevice(ctx->device));
    return BYZ_OK;
}

int read_i32(byz_ctx* ctx, const int32_t* dev, int32_t* host, int64_t count, hipStream_t stream) {
    BYZ_TRY(ctx->pinned.ensure(static_cast<size_t>(count) * sizeof(int32_t)));
    BYZ_HIP(hipMemcpyAsync(ctx->pinned.ptr, dev, static_cast<size_t>(count) * sizeof(int32_t),
                           hipMemcpyDeviceToHost, stream));
    BYZ_HIP(hipStreamSynchronize(stream));
    std::memcpy(host, ctx->pinned.ptr, static_cast<size_t>(count) * sizeof(int32_t));
    return BYZ_OK;
}

// The device-side scalars (common.hpp: layout of ctx->small) in one read-back; the sticky status word is checked and
// cleared here, so that every entry point that synchronises anyway also reports what a kernel could only flag.
int read_small(byz_ctx* ctx, int32_t (&words)[32], hipStream_t stream) {
    BYZ_TRY(read_i32(ctx, ctx->small.as<int32_t>(), words, 32, stream));
    const int32_t sticky = words[16];
    if (sticky != 0) {
        BYZ_HIP(hipMemsetAsync(device_status_word(ctx), 0, sizeof(int32_t), stream));
        if (sticky & 1) {
            set_error("gram: a K chunk never received its tile's ticket (workgroups dispatched out of order or the GPU is "
                      "shared); the distance matrix of this call is invalid");
            return BYZ_E_HIP;
        }
        if (sticky & 8) {
            set_error("krum (N <= 128): not every row's workgroup reported its score in time (GPU shared with another "
                      "process?); the result of this call is invalid");
            return BYZ_E_HIP;
        }
        if (sticky & 16) {
            set_error("column statistics: a wave of the register-resident kernel never got its turn; the statistics of this "
                      "call are invalid (please report the shape)");
            return BYZ_E_HIP;
        }
        if (sticky & 4) {
            set_error("distances: two rows have bitwise equal Gram entries but differ; their near-duplicate pairs were not "
                      "re-evaluated (please report the input)");
            return BYZ_E_UNSUPPORTED;
        }
        set_error("distances: %d near-duplicate pairs exceed the list capacity (BYZ_NEAR_PAIR_CAPACITY); their distances "
                  "were not re-evaluated", words[17]);
        return BYZ_E_UNSUPPORTED;
    }
    return BYZ_OK;
}

int ensure_distance_workspaces(byz_ctx* ctx, int64_t n) {
    BYZ_TRY(ctx->gram.ensure(static_cast<size_t>(n) * n * sizeof(double)));
    BYZ_TRY(ctx->dist.ensure(static_cast<size_t>(n) * n * sizeof(float)));
    return BYZ_OK;
}

int check_matrix(const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const char* who) {
    if (!G || n_rows <= 0 || n_cols <= 0 || ld < n_cols) {
        set_error("%s: bad matrix (ptr %p, %lld x %lld, ld %lld)", who, (const void*)G, (long long)n_rows,
                  (long long)n_cols, (long long)ld);
        return BYZ_E_INVALID;
    }
    return BYZ_OK;
}

int krum_select(byz_ctx* ctx, const float* dist, int64_t n, int64_t users_count, int64_t corrupted,
                int32_t* winner_dev, hipStream_t stream) {
    const int64_t prefix = python_prefix_len(n - 1, users_count - corrupted);
    if (krum_small_applies(n, 1))
        return launch_small_select(ctx, dist, n, prefix, nullptr, 0, 0, winner_dev, nullptr, stream);
    BYZ_TRY(launch_row_sort(ctx, dist, n, prefix, 0, false, stream));
    BYZ_TRY(launch_krum_argmin(ctx, n, winner_dev, stream));
    return BYZ_OK;
}

// krum_winner_dev != nullptr: the same row sort also forms the Krum scores of defences.py:33-34 (prefix users_count - corrupted
// of the full rows) and the Krum index goes to krum_winner_dev: configs[4]'s round runs Krum and Bulyan on ONE distance matrix,
// and a second sort of its 10,000 rows costs 7 ms
int bulyan_select(byz_ctx* ctx, const float* dist, int64_t n, int64_t users_count, int64_t corrupted,
                  int32_t* selection_dev, hipStream_t stream, int32_t* krum_winner_dev = nullptr) {
    const int64_t theta = users_count - 2 * corrupted;
    if (theta < 0 || theta > n) {
        set_error("bulyan: selection size %lld does not fit %lld rows", (long long)theta, (long long)n);
        return BYZ_E_INVALID;
    }
    // every pick scores a row by its (n_t - f) smallest of (n_t - 1) live distances (defences.py:26,34 with
    // users_count - len(selection_set) passed down from defences.py:61): the number of dropped, largest
    // entries is the same at every step
    int64_t drop = (n - 1) - users_count + corrupted;
    if (drop < 0) drop = 0;
    if (drop > n - 1) drop = n - 1;
    const int64_t krum_prefix = krum_winner_dev != nullptr ? python_prefix_len(n - 1, users_count - corrupted) : 0;
    BYZ_TRY(launch_row_sort(ctx, dist, n, krum_prefix, drop, true, stream));
    if (krum_winner_dev != nullptr) BYZ_TRY(launch_krum_argmin(ctx, n, krum_winner_dev, stream));
    int32_t* status_dev = ctx->small.as<int32_t>() + 8;
    BYZ_TRY(launch_bulyan_loop(ctx, dist, n, theta, drop, users_count, corrupted, selection_dev, status_dev, stream));
    int32_t words[32];
    BYZ_TRY(read_small(ctx, words, stream));
    const int32_t status[2] = {words[8], words[9]};
    ctx->bulyan_rescored = status[1];
    if (status[0] == 2) {
        set_error("bulyan: the selection loop's workgroups lost contact with each other (exchange timed out)");
        return BYZ_E_HIP;
    }
    if (status[0] != 0) {
        set_error("bulyan: no row scored below 1e20 (the reference raises KeyError(-1) here)");
        return BYZ_E_NO_WINNER;
    }
    return BYZ_OK;
}

}  // namespace
}  // namespace byz

using namespace byz;

extern "C" {

int byz_abi_version(void) { return BYZ_ABI_VERSION; }
const char* byz_last_error(void) { return g_error; }

int byz_ctx_create(int device, byz_ctx** out) {
    if (!out) {
        set_error("byz_ctx_create: null output");
        return BYZ_E_INVALID;
    }
    *out = nullptr;
    int count = 0;
    BYZ_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        set_error("byz_ctx_create: device %d not present (%d visible)", device, count);
        return BYZ_E_INVALID;
    }
    BYZ_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BYZ_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("libbyzagg is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
        return BYZ_E_UNSUPPORTED;
    }
    byz_ctx* ctx = new byz_ctx();
    ctx->device = device;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (ctx->small.ensure(256) != BYZ_OK || hipMemset(ctx->small.ptr, 0, 256) != hipSuccess) {
        delete ctx;
        set_error("byz_ctx_create: cannot allocate the device scalars");
        return BYZ_E_HIP;
    }
    *out = ctx;
    return BYZ_OK;
}

void byz_ctx_destroy(byz_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    byz_timing_reset(ctx);
    ctx->assemble_table.release();
    if (ctx->assemble_copied != nullptr) (void)hipEventDestroy(ctx->assemble_copied);
    ctx->gram_partials.release();
    ctx->gram.release();
    ctx->tile_order.release();
    ctx->tile_owned.release();
    ctx->gram_tickets.release();
    ctx->gram_planes.release();
    ctx->plane_order.release();
    ctx->split_redo.release();
    ctx->plane_unscale.release();
    ctx->gram_chunk_sums.release();
    ctx->dup_rep.release();
    ctx->row_signature.release();
    ctx->unique_rows.release();
    ctx->row_map.release();
    ctx->gram_compact.release();
    ctx->dist.release();
    ctx->near_pairs.release();
    ctx->near_rows.release();
    ctx->gram_rep.release();
    ctx->near_sq.release();
    ctx->near_partial.release();
    ctx->sorted_idx.release();
    ctx->large_keys.release();
    ctx->large_idx.release();
    ctx->large_rank.release();
    ctx->large_rank_t.release();
    ctx->large_dist_t.release();
    ctx->large_state.release();
    ctx->large_grid.release();
    ctx->rank_t.release();
    ctx->rank_rows.release();
    ctx->sorted_val.release();
    ctx->row_total.release();
    ctx->row_top.release();
    ctx->scores.release();
    ctx->selection.release();
    ctx->twin_class.release();
    ctx->redo_tiles.release();
    ctx->xchg.release();
    ctx->small.release();
    ctx->stage_in.release();
    ctx->stage_out.release();
    ctx->pinned.release();
    delete ctx;
}

int byz_ctx_device(const byz_ctx* ctx) { return ctx ? ctx->device : -1; }

int byz_limits(int64_t* max_rows_select, int64_t* max_rows_trimmed) {
    if (max_rows_select) *max_rows_select = select_max_rows();
    if (max_rows_trimmed) *max_rows_trimmed = trimmed_mean_max_rows();
    return BYZ_OK;
}

int byz_ctx_reserve(byz_ctx* ctx, int64_t n_rows, int64_t n_cols) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(n_rows > 0 && n_cols > 0, "byz_ctx_reserve: bad shape");
    BYZ_TRY(ensure_distance_workspaces(ctx, n_rows));
    BYZ_TRY(ctx->scores.ensure(static_cast<size_t>(n_rows) * sizeof(float)));
    BYZ_TRY(ctx->selection.ensure(static_cast<size_t>(n_rows) * sizeof(int32_t)));
    if (!select_large_applies(n_rows)) {   // (beyond 16,384 rows the selection grows its own tables: large_rows.hip)
        BYZ_TRY(ctx->sorted_idx.ensure(static_cast<size_t>(n_rows) * n_rows * sizeof(uint16_t)));
        BYZ_TRY(ctx->rank_t.ensure(static_cast<size_t>(n_rows) * n_rows * sizeof(uint16_t)));
        BYZ_TRY(ctx->rank_rows.ensure(static_cast<size_t>(n_rows) * n_rows * sizeof(uint16_t)));
    }
    BYZ_TRY(ctx->row_total.ensure(static_cast<size_t>(n_rows) * sizeof(double)));
    BYZ_TRY(ctx->row_top.ensure(static_cast<size_t>(2 * n_rows) * sizeof(double)));
    BYZ_TRY(ctx->stage_out.ensure(static_cast<size_t>(n_cols) * 3 * sizeof(float)));
    BYZ_TRY(ctx->pinned.ensure(static_cast<size_t>(n_rows) * sizeof(int32_t) + 64));
    if (krum_small_applies(n_rows, n_cols)) BYZ_TRY(reserve_small_workspaces(ctx));
    {   // the Gram's schedule words and the duplicate-row table
        const int64_t T = ceil_div(n_rows, 128);
        BYZ_TRY(ctx->gram_tickets.ensure(static_cast<size_t>(T * (T + 1) / 2 + 8) * sizeof(int32_t)));
        BYZ_TRY(ctx->dup_rep.ensure(static_cast<size_t>(n_rows + 1) * sizeof(int32_t)));
    }
    return BYZ_OK;
}

// ---- raw memory ----------------------------------------------------------------------------------
int byz_malloc(byz_ctx* ctx, int64_t bytes, void** dev_ptr) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dev_ptr && bytes > 0, "byz_malloc: bad arguments");
    BYZ_HIP(hipMalloc(dev_ptr, static_cast<size_t>(bytes)));
    return BYZ_OK;
}
int byz_free(byz_ctx* ctx, void* dev_ptr) {
    BYZ_TRY(enter(ctx));
    if (dev_ptr) BYZ_HIP(hipFree(dev_ptr));
    return BYZ_OK;
}
int byz_upload(byz_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dst_dev && src_host && bytes >= 0, "byz_upload: bad arguments");
    BYZ_HIP(hipMemcpyAsync(dst_dev, src_host, static_cast<size_t>(bytes), hipMemcpyHostToDevice, as_stream(stream)));
    return BYZ_OK;
}
int byz_download(byz_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dst_host && src_dev && bytes >= 0, "byz_download: bad arguments");
    BYZ_HIP(hipMemcpyAsync(dst_host, src_dev, static_cast<size_t>(bytes), hipMemcpyDeviceToHost, as_stream(stream)));
    BYZ_HIP(hipStreamSynchronize(as_stream(stream)));
    return BYZ_OK;
}
int byz_upload_2d(byz_ctx* ctx, void* dst_dev, int64_t dst_pitch_bytes, const void* src_host,
                  int64_t src_pitch_bytes, int64_t width_bytes, int64_t rows, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dst_dev && src_host && rows >= 0 && width_bytes >= 0, "byz_upload_2d: bad arguments");
    BYZ_HIP(hipMemcpy2DAsync(dst_dev, static_cast<size_t>(dst_pitch_bytes), src_host,
                             static_cast<size_t>(src_pitch_bytes), static_cast<size_t>(width_bytes),
                             static_cast<size_t>(rows), hipMemcpyHostToDevice, as_stream(stream)));
    return BYZ_OK;
}
int byz_stream_sync(byz_ctx* ctx, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_HIP(hipStreamSynchronize(as_stream(stream)));
    return BYZ_OK;
}

// ---- device entry points -------------------------------------------------------------------------
int byz_no_defense_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float* out,
                       void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "no_defense"));
    return launch_column_mean(ctx, G, n_rows, n_cols, ld, out, as_stream(stream));
}

int byz_gram_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, double* gram,
                 void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "gram"));
    return launch_gram(ctx, G, n_rows, n_cols, ld, gram, as_stream(stream));
}

int byz_gram_share_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                       const int32_t* row_index, int share_count, int share_index, double* gram, void* stream) {
    BYZ_TRY(enter(ctx));
    // with row_index the logical matrix has n_rows rows taken from anywhere in G: only the pointer and ld are checkable
    BYZ_REQUIRE(G && n_rows > 0 && n_cols > 0 && ld >= n_cols, "gram_share: bad matrix");
    return launch_gram_share(ctx, G, n_rows, n_cols, ld, row_index, share_count, share_index, gram, as_stream(stream));
}

int byz_gram_share_add_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                           const int32_t* row_index, int share_count, int share_index, double* gram, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(G && n_rows > 0 && n_cols > 0 && ld >= n_cols, "gram_share_add: bad matrix");
    return launch_gram_share(ctx, G, n_rows, n_cols, ld, row_index, share_count, share_index, gram, as_stream(stream), true);
}

int byz_distances_from_gram_dev(byz_ctx* ctx, const double* gram, int64_t n_rows, float* dist, void* stream) {
    BYZ_TRY(enter(ctx));
    ctx->row_map_rows = 0;   // an external Gram: nothing is known about its rows
    return launch_distances_from_gram(ctx, gram, n_rows, dist, as_stream(stream), nullptr, 0, 0);
}

int byz_near_pairs_count(byz_ctx* ctx, int64_t* count_host, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(count_host, "near_pairs_count: null output");
    int32_t words[32];
    BYZ_TRY(read_small(ctx, words, as_stream(stream)));
    *count_host = words[17];
    if (words[17] > ctx->near_pair_capacity) {
        set_error("distances: %d near-duplicate pairs exceed the list capacity %lld (BYZ_NEAR_PAIR_CAPACITY)", words[17],
                  (long long)ctx->near_pair_capacity);
        return BYZ_E_UNSUPPORTED;
    }
    return BYZ_OK;
}

int byz_near_pairs_sqdist_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                              const int32_t* row_index, double* sq_dev, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "near_pairs_sqdist"));
    BYZ_REQUIRE(sq_dev && ctx->near_pair_capacity > 0, "near_pairs_sqdist: no pair list (call byz_distances_from_gram_dev first)");
    return launch_near_pair_sqdist(ctx, G, n_cols, ld, row_index, sq_dev, as_stream(stream));
}

int byz_near_pairs_apply_dev(byz_ctx* ctx, const double* sq_dev, int64_t n_rows, float* dist, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(sq_dev && dist && n_rows > 0 && ctx->near_pair_capacity > 0, "near_pairs_apply: bad arguments");
    return launch_near_pair_apply(ctx, sq_dev, n_rows, dist, as_stream(stream));
}

int byz_ctx_check(byz_ctx* ctx, void* stream) {
    BYZ_TRY(enter(ctx));
    int32_t words[32];
    return read_small(ctx, words, as_stream(stream));
}

int byz_bulyan_rescored(const byz_ctx* ctx, int64_t* rows_host) {
    if (!ctx || !rows_host) {
        set_error("bulyan_rescored: null argument");
        return BYZ_E_INVALID;
    }
    *rows_host = ctx->bulyan_rescored;
    return BYZ_OK;
}

int byz_pairwise_distances_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                               float* dist, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "pairwise_distances"));
    BYZ_REQUIRE(dist, "pairwise_distances: null output");
    if (krum_small_applies(n_rows, n_cols)) {
        ctx->row_map_rows = 0;
        return launch_small_distances(ctx, G, n_rows, n_cols, ld, dist, as_stream(stream));
    }
    BYZ_TRY(ctx->gram.ensure(static_cast<size_t>(n_rows) * n_rows * sizeof(double)));
    BYZ_TRY(launch_gram(ctx, G, n_rows, n_cols, ld, ctx->gram.as<double>(), as_stream(stream)));
    return launch_distances_from_gram(ctx, ctx->gram.as<double>(), n_rows, dist, as_stream(stream), G, n_cols, ld);
}

int byz_krum_select_dev(byz_ctx* ctx, const float* dist, int64_t n_rows, int64_t users_count,
                        int64_t corrupted_count, int32_t* index_host, float* scores_dev, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dist && n_rows > 0, "krum_select: bad arguments");
    hipStream_t s = as_stream(stream);
    BYZ_TRY(krum_select(ctx, dist, n_rows, users_count, corrupted_count, ctx->small.as<int32_t>(), s));
    if (scores_dev)
        BYZ_HIP(hipMemcpyAsync(scores_dev, ctx->scores.ptr, static_cast<size_t>(n_rows) * sizeof(float),
                               hipMemcpyDeviceToDevice, s));
    if (index_host) {
        int32_t words[32];
        BYZ_TRY(read_small(ctx, words, s));
        *index_host = words[0];
    }
    return BYZ_OK;
}

int byz_krum_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t users_count,
                 int64_t corrupted_count, int check_assert, float* out_row, int32_t* index_host, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "krum"));
    if (check_assert && !(users_count >= 2 * corrupted_count + 1)) {  // defences.py:25
        set_error("('users_count>=2*corrupted_count + 3', %lld, %lld)", (long long)users_count,
                  (long long)corrupted_count);
        return BYZ_E_PRECONDITION;
    }
    hipStream_t s = as_stream(stream);
    BYZ_TRY(ensure_distance_workspaces(ctx, n_rows));
    int32_t* winner = ctx->small.as<int32_t>();
    if (krum_small_applies(n_rows, n_cols)) {
        // the reference's own sizes: two launches (krum_small.hip), the row copy is part of the second
        ctx->row_map_rows = 0;
        const int64_t prefix = python_prefix_len(n_rows - 1, users_count - corrupted_count);
        if (ctx->num_cus >= n_rows) {
            BYZ_TRY(launch_small_krum(ctx, G, n_rows, n_cols, ld, ctx->dist.as<float>(), prefix, winner, out_row, s));
        } else {
            // the two-launch form has every row's workgroup wait for all the others' scores: they must all be resident at once
            // (one workgroup per CU).  On a device or partition with fewer CUs than rows (a CPX partition has 32) the first wave
            // of workgroups would spin to the limit and the call fail with the status word set: take the
            // four-launch form, which has no wait across workgroups.
            BYZ_TRY(launch_small_distances(ctx, G, n_rows, n_cols, ld, ctx->dist.as<float>(), s));
            BYZ_TRY(launch_small_select(ctx, ctx->dist.as<float>(), n_rows, prefix, G, n_cols, ld, winner, out_row, s));
        }
    } else {
        BYZ_TRY(launch_gram(ctx, G, n_rows, n_cols, ld, ctx->gram.as<double>(), s));
        BYZ_TRY(launch_distances_from_gram(ctx, ctx->gram.as<double>(), n_rows, ctx->dist.as<float>(), s, G, n_cols, ld));
        BYZ_TRY(krum_select(ctx, ctx->dist.as<float>(), n_rows, users_count, corrupted_count, winner, s));
        if (out_row) BYZ_TRY(launch_copy_row(ctx, G, ld, n_rows, n_cols, winner, out_row, s));
    }
    if (index_host) {
        int32_t words[32];
        BYZ_TRY(read_small(ctx, words, s));
        *index_host = words[0];
    }
    return BYZ_OK;
}

int byz_trimmed_mean_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                         const int32_t* row_index, int64_t corrupted_count, float* out, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "trimmed_mean"));
    // defences.py:45: number_to_consider = int(rows - corrupted) - 1, then a Python slice [:k]
    const int64_t keep = python_prefix_len(n_rows, n_rows - corrupted_count - 1);
    return launch_trimmed_mean(ctx, G, n_rows, n_cols, ld, row_index, keep, out, as_stream(stream));
}

int byz_trimmed_mean_redone(byz_ctx* ctx, int64_t* tiles_host, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(tiles_host, "trimmed_mean_redone: null output");
    *tiles_host = 0;
    if (ctx->redo_tiles.ptr == nullptr || !ctx->redo_valid) return BYZ_OK;
    int32_t count = 0;
    BYZ_TRY(read_i32(ctx, ctx->redo_tiles.as<int32_t>(), &count, 1, as_stream(stream)));
    *tiles_host = count;
    return BYZ_OK;
}

int byz_bulyan_select_dev(byz_ctx* ctx, const float* dist, int64_t n_rows, int64_t users_count,
                          int64_t corrupted_count, int32_t* selection, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dist && selection && n_rows > 0, "bulyan_select: bad arguments");
    return bulyan_select(ctx, dist, n_rows, users_count, corrupted_count, selection, as_stream(stream));
}

int byz_krum_bulyan_select_dev(byz_ctx* ctx, const float* dist, int64_t n_rows, int64_t users_count,
                               int64_t corrupted_count, int32_t* krum_index_host, int32_t* selection, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dist && selection && krum_index_host && n_rows > 0, "krum_bulyan_select: bad arguments");
    hipStream_t s = as_stream(stream);
    BYZ_TRY(bulyan_select(ctx, dist, n_rows, users_count, corrupted_count, selection, s, ctx->small.as<int32_t>()));
    int32_t words[32];
    BYZ_TRY(read_small(ctx, words, s));
    *krum_index_host = words[0];
    return BYZ_OK;
}

int byz_bulyan_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t users_count,
                   int64_t corrupted_count, float* out, int32_t* selection_out, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "bulyan"));
    BYZ_REQUIRE(out, "bulyan: null output");
    if (!(users_count >= 4 * corrupted_count + 3)) {  // defences.py:56
        set_error("bulyan: users_count >= 4*corrupted_count + 3 violated (%lld, %lld)", (long long)users_count,
                  (long long)corrupted_count);
        return BYZ_E_PRECONDITION;
    }
    hipStream_t s = as_stream(stream);
    const int64_t theta = users_count - 2 * corrupted_count;
    BYZ_TRY(ensure_distance_workspaces(ctx, n_rows));
    BYZ_TRY(ctx->selection.ensure(static_cast<size_t>(n_rows) * sizeof(int32_t)));
    if (krum_small_applies(n_rows, n_cols)) {
        ctx->row_map_rows = 0;
        BYZ_TRY(launch_small_distances(ctx, G, n_rows, n_cols, ld, ctx->dist.as<float>(), s));
    } else {
        BYZ_TRY(launch_gram(ctx, G, n_rows, n_cols, ld, ctx->gram.as<double>(), s));
        BYZ_TRY(launch_distances_from_gram(ctx, ctx->gram.as<double>(), n_rows, ctx->dist.as<float>(), s, G, n_cols, ld));
    }
    int32_t* sel = ctx->selection.as<int32_t>();
    BYZ_TRY(bulyan_select(ctx, ctx->dist.as<float>(), n_rows, users_count, corrupted_count, sel, s));
    if (selection_out)
        BYZ_HIP(hipMemcpyAsync(selection_out, sel, static_cast<size_t>(theta) * sizeof(int32_t),
                               hipMemcpyDeviceToDevice, s));
    // defences.py:70: trimmed_mean(np.array(selection_set), len(selection_set), 2*corrupted_count)
    const int64_t keep = python_prefix_len(theta, theta - 2 * corrupted_count - 1);
    return launch_trimmed_mean(ctx, G, theta, n_cols, ld, sel, keep, out, s);
}

// ---- multi-GPU, columns layout: the one exchange of the path through the host's all-reduce -------------------
namespace {

int reduce_over_ranks(byz_allreduce_f64_fn allreduce, void* user, double* buf, int64_t count, void* stream, const char* what) {
    const int rc = allreduce(user, buf, count, stream);
    if (rc != 0) {
        set_error("%s: the caller's all-reduce returned %d", what, rc);
        return BYZ_E_COLLECTIVE;
    }
    return BYZ_OK;
}

// dist (n x n) from the column slices of all ranks: Gram of the local slice -> all-reduce -> distances -> the near-duplicate
// pairs of the all-reduced Gram re-evaluated on the difference (per-rank sums over the local columns -> all-reduce -> apply).
// The sequence of sharded.py's global_distances + engine.distances_from_gram, in one call.
int sharded_distances(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, byz_allreduce_f64_fn allreduce,
                      void* user, float* dist, void* stream) {
    hipStream_t s = as_stream(stream);
    BYZ_TRY(ctx->gram.ensure(static_cast<size_t>(n_rows) * n_rows * sizeof(double)));
    double* gram = ctx->gram.as<double>();
    BYZ_TRY(launch_gram(ctx, G, n_rows, n_cols, ld, gram, s));
    // rows that are identical in THIS slice need not be identical in the others: what launch_gram's duplicate search proved
    // holds for the local columns only, and the all-reduced Gram is an external one as far as the distances are concerned
    ctx->row_map_rows = 0;
    BYZ_TRY(reduce_over_ranks(allreduce, user, gram, n_rows * n_rows, stream, "sharded distances (Gram)"));
    BYZ_TRY(launch_distances_from_gram(ctx, gram, n_rows, dist, s, nullptr, 0, 0));
    int32_t words[32];
    BYZ_TRY(read_small(ctx, words, s));   // (the same all-reduced Gram on every rank: the same count, the same list)
    const int64_t count = words[17];
    if (count > ctx->near_pair_capacity) {
        set_error("distances: %lld near-duplicate pairs exceed the list capacity %lld (BYZ_NEAR_PAIR_CAPACITY)", (long long)count,
                  (long long)ctx->near_pair_capacity);
        return BYZ_E_UNSUPPORTED;
    }
    if (count > 0) {
        double* sq = ctx->near_sq.as<double>();
        BYZ_TRY(launch_near_pair_sqdist(ctx, G, n_cols, ld, nullptr, sq, s));
        BYZ_TRY(reduce_over_ranks(allreduce, user, sq, count, stream, "sharded distances (near pairs)"));
        BYZ_TRY(launch_near_pair_apply(ctx, sq, n_rows, dist, s));
    }
    return BYZ_OK;
}

}  // namespace

int byz_pairwise_distances_sharded_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld,
                                       byz_allreduce_f64_fn allreduce, void* user, float* dist, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "pairwise_distances_sharded"));
    BYZ_REQUIRE(dist && allreduce, "pairwise_distances_sharded: null output or null all-reduce");
    return sharded_distances(ctx, G, n_rows, n_cols, ld, allreduce, user, dist, stream);
}

int byz_krum_sharded_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t users_count,
                         int64_t corrupted_count, int check_assert, byz_allreduce_f64_fn allreduce, void* user, float* out_row,
                         int32_t* index_host, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "krum_sharded"));
    BYZ_REQUIRE(allreduce, "krum_sharded: null all-reduce");
    if (check_assert && !(users_count >= 2 * corrupted_count + 1)) {  // defences.py:25
        set_error("('users_count>=2*corrupted_count + 3', %lld, %lld)", (long long)users_count, (long long)corrupted_count);
        return BYZ_E_PRECONDITION;
    }
    hipStream_t s = as_stream(stream);
    BYZ_TRY(ensure_distance_workspaces(ctx, n_rows));
    BYZ_TRY(sharded_distances(ctx, G, n_rows, n_cols, ld, allreduce, user, ctx->dist.as<float>(), stream));
    int32_t* winner = ctx->small.as<int32_t>();
    BYZ_TRY(krum_select(ctx, ctx->dist.as<float>(), n_rows, users_count, corrupted_count, winner, s));
    if (out_row) BYZ_TRY(launch_copy_row(ctx, G, ld, n_rows, n_cols, winner, out_row, s));
    if (index_host) {
        int32_t words[32];
        BYZ_TRY(read_small(ctx, words, s));
        *index_host = words[0];
    }
    return BYZ_OK;
}

int byz_bulyan_sharded_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t users_count,
                           int64_t corrupted_count, byz_allreduce_f64_fn allreduce, void* user, float* out,
                           int32_t* selection_out, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "bulyan_sharded"));
    BYZ_REQUIRE(out && allreduce, "bulyan_sharded: null output or null all-reduce");
    if (!(users_count >= 4 * corrupted_count + 3)) {  // defences.py:56
        set_error("bulyan: users_count >= 4*corrupted_count + 3 violated (%lld, %lld)", (long long)users_count,
                  (long long)corrupted_count);
        return BYZ_E_PRECONDITION;
    }
    hipStream_t s = as_stream(stream);
    const int64_t theta = users_count - 2 * corrupted_count;
    BYZ_TRY(ensure_distance_workspaces(ctx, n_rows));
    BYZ_TRY(ctx->selection.ensure(static_cast<size_t>(n_rows) * sizeof(int32_t)));
    BYZ_TRY(sharded_distances(ctx, G, n_rows, n_cols, ld, allreduce, user, ctx->dist.as<float>(), stream));
    int32_t* sel = ctx->selection.as<int32_t>();
    BYZ_TRY(bulyan_select(ctx, ctx->dist.as<float>(), n_rows, users_count, corrupted_count, sel, s));
    if (selection_out)
        BYZ_HIP(hipMemcpyAsync(selection_out, sel, static_cast<size_t>(theta) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    // defences.py:70 on the local columns of the selected rows
    const int64_t keep = python_prefix_len(theta, theta - 2 * corrupted_count - 1);
    return launch_trimmed_mean(ctx, G, theta, n_cols, ld, sel, keep, out, s);
}

int byz_drift_attack_dev(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, float num_std,
                         float* drift, float* mean, float* stdev, int write_back, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "drift_attack"));
    hipStream_t s = as_stream(stream);
    float* vec = drift;
    if (write_back && !vec) {
        BYZ_TRY(ctx->stage_out.ensure(static_cast<size_t>(n_cols) * 3 * sizeof(float)));
        vec = ctx->stage_out.as<float>();
    }
    BYZ_TRY(launch_column_drift(ctx, G, n_rows, n_cols, ld, num_std, vec, mean, stdev, s));
    if (write_back) BYZ_TRY(launch_broadcast_rows(ctx, G, n_rows, n_cols, ld, vec, s));
    return BYZ_OK;
}

int byz_column_chain_dev(byz_ctx* ctx, const float* G, int64_t n_rows, int64_t n_cols, int64_t ld, const float* carry_in,
                         const float* mean, float* out_sum, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "column_chain"));
    BYZ_REQUIRE(out_sum, "column_chain: null output");
    return launch_column_chain(ctx, G, n_rows, n_cols, ld, carry_in, mean, out_sum, as_stream(stream));
}

int byz_column_finish_dev(byz_ctx* ctx, const float* sum, const float* sumsq, int64_t total_rows, float num_std, int64_t n_cols,
                          float* mean, float* stdev, float* drift, void* stream) {
    BYZ_TRY(enter(ctx));
    return launch_column_finish(ctx, sum, sumsq, total_rows, num_std, n_cols, mean, stdev, drift, as_stream(stream));
}

int byz_drift_axpy_dev(byz_ctx* ctx, float* mean, const float* stdev, int64_t n, float num_std, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(mean && stdev && n > 0, "drift_axpy: bad arguments");
    return launch_drift_axpy(ctx, mean, stdev, n, num_std, as_stream(stream));
}

int byz_server_update_dev(byz_ctx* ctx, float* weights, float* velocity, const float* agg, int64_t n,
                          float momentum, float learning_rate, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(weights && velocity && agg && n > 0, "server_update: bad arguments");
    return launch_server_update(ctx, weights, velocity, agg, n, momentum, learning_rate, as_stream(stream));
}

// ---- the steps either side of the path -------------------------------------------
int byz_backdoor_initial_params_dev(byz_ctx* ctx, const float* original_params, const float* grads_mean, int64_t n,
                                    float learning_rate, float* out, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(original_params && grads_mean && out && n > 0, "backdoor_initial_params: bad arguments");
    return launch_backdoor_initial(ctx, original_params, grads_mean, n, learning_rate, out, as_stream(stream));
}

int byz_backdoor_clip_dev(byz_ctx* ctx, const float* grads_mean, const float* grads_stdev, const float* original_params,
                          const float* mal_net_params, int64_t n, float learning_rate, float num_std, float* out,
                          void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(grads_mean && grads_stdev && original_params && mal_net_params && out && n > 0,
                "backdoor_clip: bad arguments");
    return launch_backdoor_clip(ctx, grads_mean, grads_stdev, original_params, mal_net_params, n, learning_rate,
                                num_std, out, as_stream(stream));
}

int byz_assemble_row_dev(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t row,
                         int64_t n_segments, const float* const* segments_dev, const int64_t* lengths, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "assemble_row"));
    BYZ_REQUIRE(row >= 0 && row < n_rows, "assemble_row: row %lld outside 0..%lld", (long long)row, (long long)n_rows - 1);
    BYZ_REQUIRE(n_segments > 0 && segments_dev && lengths, "assemble_row: no segments");
    return launch_assemble_row(ctx, G + row * ld, n_cols, n_segments, segments_dev, lengths, as_stream(stream));
}

int byz_assemble_rows_dev(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t first_row,
                          int64_t n_clients, int64_t n_segments, const float* const* segments_dev, const int64_t* lengths,
                          void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "assemble_rows"));
    BYZ_REQUIRE(first_row >= 0 && n_clients > 0 && first_row + n_clients <= n_rows,
                "assemble_rows: rows %lld .. %lld outside 0..%lld", (long long)first_row, (long long)(first_row + n_clients - 1),
                (long long)n_rows - 1);
    BYZ_REQUIRE(n_segments > 0 && segments_dev && lengths, "assemble_rows: no segments");
    return launch_assemble_rows(ctx, G + first_row * ld, n_cols, ld, n_clients, n_segments, segments_dev, lengths,
                                as_stream(stream));
}

int byz_assemble_rows_again_dev(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t first_row,
                                int64_t n_clients, int64_t n_segments, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "assemble_rows_again"));
    BYZ_REQUIRE(first_row >= 0 && n_clients > 0 && first_row + n_clients <= n_rows,
                "assemble_rows_again: rows %lld .. %lld outside 0..%lld", (long long)first_row,
                (long long)(first_row + n_clients - 1), (long long)n_rows - 1);
    return launch_assemble_rows_again(ctx, G + first_row * ld, n_cols, ld, n_clients, n_segments, as_stream(stream));
}

int byz_assemble_columns_dev(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t n_segments,
                             const float* const* segments_dev, const int64_t* lengths, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "assemble_columns"));
    BYZ_REQUIRE(n_segments > 0 && segments_dev && lengths, "assemble_columns: no segments");
    return launch_assemble_columns(ctx, G, n_rows, n_cols, ld, n_segments, segments_dev, lengths, as_stream(stream));
}

int byz_assemble_row_host(byz_ctx* ctx, float* G, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t row,
                          const float* grads_host, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G, n_rows, n_cols, ld, "assemble_row"));
    BYZ_REQUIRE(row >= 0 && row < n_rows && grads_host, "assemble_row: row %lld outside 0..%lld", (long long)row,
                (long long)n_rows - 1);
    BYZ_HIP(hipMemcpyAsync(G + row * ld, grads_host, static_cast<size_t>(n_cols) * sizeof(float), hipMemcpyHostToDevice,
                           as_stream(stream)));
    return BYZ_OK;
}

// ---- host-pointer convenience --------------------------------------------------------------------
int byz_defend_host(byz_ctx* ctx, int name, const float* G_host, int64_t n_rows, int64_t n_cols,
                    int64_t users_count, int64_t corrupted_count, int check_assert, float* out_host,
                    int32_t* aux_host) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G_host, n_rows, n_cols, n_cols, "defend"));
    BYZ_REQUIRE(name >= 0 && name <= 3, "defend: unknown defence %d", name);
    // preconditions first: the reference asserts before touching the data
    if (name == 1 && check_assert && !(users_count >= 2 * corrupted_count + 1)) {
        set_error("('users_count>=2*corrupted_count + 3', %lld, %lld)", (long long)users_count, (long long)corrupted_count);
        return BYZ_E_PRECONDITION;
    }
    if (name == 3 && !(users_count >= 4 * corrupted_count + 3)) {
        set_error("bulyan: users_count >= 4*corrupted_count + 3 violated (%lld, %lld)", (long long)users_count, (long long)corrupted_count);
        return BYZ_E_PRECONDITION;
    }
    hipStream_t s = nullptr;
    const size_t bytes = static_cast<size_t>(n_rows) * n_cols * sizeof(float);
    BYZ_TRY(ctx->stage_in.ensure(bytes));
    BYZ_TRY(ctx->stage_out.ensure(static_cast<size_t>(n_cols) * 3 * sizeof(float)));
    float* G = ctx->stage_in.as<float>();
    float* out = ctx->stage_out.as<float>();
    BYZ_HIP(hipMemcpyAsync(G, G_host, bytes, hipMemcpyHostToDevice, s));
    int32_t index = 0;
    switch (name) {
        case 0: BYZ_TRY(byz_no_defense_dev(ctx, G, n_rows, n_cols, n_cols, out, s)); break;
        case 1:
            BYZ_TRY(byz_krum_dev(ctx, G, n_rows, n_cols, n_cols, users_count, corrupted_count, 0,
                                 out_host ? out : nullptr, &index, s));
            if (aux_host) aux_host[0] = index;
            break;
        case 2: BYZ_TRY(byz_trimmed_mean_dev(ctx, G, n_rows, n_cols, n_cols, nullptr, corrupted_count, out, s)); break;
        case 3: {
            BYZ_TRY(byz_bulyan_dev(ctx, G, n_rows, n_cols, n_cols, users_count, corrupted_count, out, nullptr, s));
            if (aux_host) {
                const int64_t theta = users_count - 2 * corrupted_count;
                BYZ_TRY(read_i32(ctx, ctx->selection.as<int32_t>(), aux_host, theta, s));
            }
            break;
        }
    }
    if (out_host) {
        BYZ_HIP(hipMemcpyAsync(out_host, out, static_cast<size_t>(n_cols) * sizeof(float), hipMemcpyDeviceToHost, s));
        BYZ_HIP(hipStreamSynchronize(s));
    }
    return BYZ_OK;
}

int byz_pairwise_distances_host(byz_ctx* ctx, const float* G_host, int64_t n_rows, int64_t n_cols,
                                float* dist_host) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(G_host, n_rows, n_cols, n_cols, "pairwise_distances"));
    BYZ_REQUIRE(dist_host, "pairwise_distances: null output");
    hipStream_t s = nullptr;
    const size_t bytes = static_cast<size_t>(n_rows) * n_cols * sizeof(float);
    BYZ_TRY(ctx->stage_in.ensure(bytes));
    BYZ_TRY(ensure_distance_workspaces(ctx, n_rows));
    BYZ_HIP(hipMemcpyAsync(ctx->stage_in.ptr, G_host, bytes, hipMemcpyHostToDevice, s));
    BYZ_TRY(byz_pairwise_distances_dev(ctx, ctx->stage_in.as<float>(), n_rows, n_cols, n_cols,
                                       ctx->dist.as<float>(), s));
    BYZ_HIP(hipMemcpyAsync(dist_host, ctx->dist.ptr, static_cast<size_t>(n_rows) * n_rows * sizeof(float),
                           hipMemcpyDeviceToHost, s));
    BYZ_HIP(hipStreamSynchronize(s));
    return BYZ_OK;
}

int byz_krum_select_host(byz_ctx* ctx, const float* dist_host, int64_t n_rows, int64_t users_count,
                         int64_t corrupted_count, int32_t* index_host) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(dist_host && index_host && n_rows > 0, "krum_select: bad arguments");
    hipStream_t s = nullptr;
    BYZ_TRY(ensure_distance_workspaces(ctx, n_rows));
    BYZ_HIP(hipMemcpyAsync(ctx->dist.ptr, dist_host, static_cast<size_t>(n_rows) * n_rows * sizeof(float),
                           hipMemcpyHostToDevice, s));
    return byz_krum_select_dev(ctx, ctx->dist.as<float>(), n_rows, users_count, corrupted_count, index_host,
                               nullptr, s);
}

int byz_drift_attack_host(byz_ctx* ctx, const float* rows_host, int64_t n_rows, int64_t n_cols, float num_std,
                          float* drift_host, float* mean_host, float* std_host) {
    BYZ_TRY(enter(ctx));
    BYZ_TRY(check_matrix(rows_host, n_rows, n_cols, n_cols, "drift_attack"));
    hipStream_t s = nullptr;
    const size_t bytes = static_cast<size_t>(n_rows) * n_cols * sizeof(float);
    const size_t vec = static_cast<size_t>(n_cols) * sizeof(float);
    BYZ_TRY(ctx->stage_in.ensure(bytes));
    BYZ_TRY(ctx->stage_out.ensure(3 * vec));
    float* out = ctx->stage_out.as<float>();
    BYZ_HIP(hipMemcpyAsync(ctx->stage_in.ptr, rows_host, bytes, hipMemcpyHostToDevice, s));
    BYZ_TRY(launch_column_drift(ctx, ctx->stage_in.as<float>(), n_rows, n_cols, n_cols, num_std, out, out + n_cols,
                                out + 2 * n_cols, s));
    if (drift_host) BYZ_HIP(hipMemcpyAsync(drift_host, out, vec, hipMemcpyDeviceToHost, s));
    if (mean_host) BYZ_HIP(hipMemcpyAsync(mean_host, out + n_cols, vec, hipMemcpyDeviceToHost, s));
    if (std_host) BYZ_HIP(hipMemcpyAsync(std_host, out + 2 * n_cols, vec, hipMemcpyDeviceToHost, s));
    int32_t words[32];
    return read_small(ctx, words, s);     // synchronises; a kernel that flagged a failure makes this call fail
}

// ---- timing --------------------------------------------------------------------------------------
static const char* const kKernelNames[BYZ_K_COUNT] = {
    "column_stats", "gram_tile", "gram_reduce", "distances", "row_sort",
    "krum_argmin",  "bulyan_loop", "trimmed_mean", "misc", "plane_split"};

const char* byz_kernel_name(int kernel) {
    return (kernel >= 0 && kernel < BYZ_K_COUNT) ? kKernelNames[kernel] : "?";
}

int byz_timing_enable(byz_ctx* ctx, int on) {
    BYZ_TRY(enter(ctx));
    ctx->timing = on != 0;
    return BYZ_OK;
}

int byz_timing_reset(byz_ctx* ctx) {
    if (!ctx) return BYZ_E_INVALID;
    for (auto& slot : ctx->slots) {
        for (auto& ev : slot.pending) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        slot.pending.clear();
        slot.total_ms = 0.0;
        slot.launches = 0;
    }
    return BYZ_OK;
}

int byz_timing_read(byz_ctx* ctx, int kernel, double* total_ms, int64_t* launches) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(kernel >= 0 && kernel < BYZ_K_COUNT, "timing: bad kernel id");
    auto& slot = ctx->slots[kernel];
    for (auto& ev : slot.pending) {
        BYZ_HIP(hipEventSynchronize(ev.second));
        float ms = 0.0f;
        BYZ_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        slot.total_ms += ms;
        slot.launches += 1;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    slot.pending.clear();
    if (total_ms) *total_ms = slot.total_ms;
    if (launches) *launches = slot.launches;
    return BYZ_OK;
}

int byz_selftest_lane_exchange_dev(byz_ctx* ctx, int32_t* out_dev, int32_t* n_patterns_host, void* stream) {
    BYZ_TRY(enter(ctx));
    BYZ_REQUIRE(out_dev && n_patterns_host, "selftest: bad arguments");
    return launch_lane_selftest(ctx, out_dev, n_patterns_host, as_stream(stream));
}

}  // extern "C"
