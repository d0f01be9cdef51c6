/*
 * byzagg -- C ABI of the MI355X (gfx950) Byzantine-robust aggregation engine.
 *
 * This is the drop-in boundary for the hot path of shaneson0/attacking_federate_learning:
 * what `defences.py` and `malicious.py` compute on the host with numpy, this library
 * computes on one GPU with hand-written HIP kernels.  Plain C types only: pointers, sizes,
 * a stream handle passed as `void*` (a `hipStream_t`, NULL = the default stream).  No torch
 * types cross this line.  The reference is pure Python, so a maintainer binds these with
 * ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - G is the (n_rows x n_cols) fp32 gradient matrix, row-major, leading dimension `ld`
 *     (elements).  It corresponds to `Server.users_grads` (reference server.py:34-35).
 *   - "_dev" pointers are device pointers; "_host" entry points take host pointers and do the
 *     staging copies themselves (pinned bounce buffers owned by the context).
 *   - Every call is asynchronous on `stream` unless it has a host output, in which case it
 *     returns after that output is valid.
 *   - Return value: BYZ_OK (0) or a negative BYZ_E_* code; `byz_last_error()` has the text.
 *     BYZ_E_PRECONDITION mirrors the reference's `assert`s (defences.py:25, 56): the Python
 *     shim turns it into AssertionError.
 *   - The context owns all workspaces (distance matrix, sort buffers, split-K slabs).  They
 *     grow on demand outside the kernels; `byz_ctx_reserve` pre-sizes them so that no
 *     allocation happens on the hot path.
 *   - One call in flight per context (the reference is single-threaded and synchronous).
 */
#ifndef BYZAGG_H
#define BYZAGG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BYZ_ABI_VERSION 1

enum {
    BYZ_OK = 0,
    BYZ_E_INVALID = -1,      /* bad argument (null pointer, negative size, ld < n_cols ...)   */
    BYZ_E_PRECONDITION = -2, /* the reference would raise AssertionError                       */
    BYZ_E_HIP = -3,          /* a HIP runtime call failed                                      */
    BYZ_E_UNSUPPORTED = -4,  /* size beyond what this build supports (see byz_limits)          */
    BYZ_E_NO_WINNER = -5,    /* Krum found no score < 1e20 (reference: index -1 / KeyError)    */
    BYZ_E_COLLECTIVE = -6    /* the caller's all-reduce (byz_allreduce_f64_fn) reported failure */
};

typedef struct byz_ctx byz_ctx;

/* ---- context -------------------------------------------------------------------------- */
int byz_abi_version(void);
const char* byz_last_error(void);                       /* thread-local text of the last failure */
int byz_ctx_create(int device, byz_ctx** out);
void byz_ctx_destroy(byz_ctx* ctx);
int byz_ctx_reserve(byz_ctx* ctx, int64_t n_rows, int64_t n_cols);
int byz_ctx_device(const byz_ctx* ctx);
/* Synchronises `stream` and reports what a kernel of an asynchronous call could only flag on the device:  */
/* BYZ_E_HIP when a Gram chunk never got its accumulation ticket (the distances of that call are invalid), */
/* BYZ_E_UNSUPPORTED when the near-duplicate pair list overflowed.  Entry points with a host output do this */
/* themselves.                                                                                             */
int byz_ctx_check(byz_ctx* ctx, void* stream);
/* largest supported row count for the selection kernels (rows of G: Krum, Bulyan) and for trimmed_mean: */
/* 2^20 both -- an index width, not a kernel's capacity: the reference has no limit (defences.py:23-70),   */
/* and beyond the 16,384 rows the LDS-resident kernels hold (BASELINE's largest configuration has 10,000   */
/* clients) the rows are sorted in global memory and the Bulyan loop runs in batches of picks              */
/* (csrc/large_rows.hip: the same results; Bulyan's selection of 20,000 rows: 0.9 s); memory -- 24 N^2     */
/* bytes of tables -- is what ends it in practice.  Beyond 2^20 the calls return BYZ_E_UNSUPPORTED.        */
/* trimmed_mean runs its fast kernels up to 5376 rows.                                                     */
int byz_limits(int64_t* max_rows_select, int64_t* max_rows_trimmed);

/* ---- raw device memory, so that a host without torch can still drive the library ------- */
int byz_malloc(byz_ctx* ctx, int64_t bytes, void** dev_ptr);
int byz_free(byz_ctx* ctx, void* dev_ptr);
int byz_upload(byz_ctx* ctx, void* dst_dev, const void* src_host, int64_t bytes, void* stream);
int byz_download(byz_ctx* ctx, void* dst_host, const void* src_dev, int64_t bytes, void* stream);
int byz_upload_2d(byz_ctx* ctx, void* dst_dev, int64_t dst_pitch_bytes, const void* src_host,
                  int64_t src_pitch_bytes, int64_t width_bytes, int64_t rows, void* stream);
int byz_stream_sync(byz_ctx* ctx, void* stream);

/* ---- defences.no_defense (reference defences.py:13-14) --------------------------------- */
/* out_dev[c] = mean over rows of G[:, c].                                                  */
int byz_no_defense_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols,
                       int64_t ld, float* out_dev, void* stream);

/* ---- defences._krum_create_distances (reference defences.py:16-21) --------------------- */
/* dist_dev: (n_rows x n_rows) fp32, row-major, unsquared L2 distances, diagonal = +inf     */
/* (the reference stores no self-distance).  Gram via fp32 MFMA, chunk partials in fp64.    */
int byz_pairwise_distances_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols,
                               int64_t ld, float* dist_dev, void* stream);
/* The two halves of the above, exposed for the D-sharded multi-GPU path: every rank computes */
/* the Gram of its column slice in fp64, the host all-reduces it, then every rank converts.   */
int byz_gram_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                 double* gram_dev, void* stream);
/* One rank's SHARE of a Gram, for ranks that all hold the same rows (the client-sharded path after an   */
/* all-gather of a column panel): every share_count-th 128 x 128 tile of the lower triangle starting with */
/* share_index is computed, the rest of gram_dev is written as zero, so that the SUM over the ranks is    */
/* the panel's Gram.  row_index_dev (optional): logical row r is G[row_index[r]] (skips padding rows of   */
/* the gathered panel).                                                                                    */
int byz_gram_share_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                       const int32_t* row_index_dev, int share_count, int share_index, double* gram_dev,
                       void* stream);
/* The same share ADDED into gram_dev (which the caller zeroed before the first panel): the sum over column  */
/* panels accumulates in the caller's one N x N buffer, in panel order, instead of through a separate N x N    */
/* addition pass per panel; entries of other ranks' tiles are left untouched.                                 */
int byz_gram_share_add_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                           const int32_t* row_index_dev, int share_count, int share_index, double* gram_dev,
                           void* stream);
int byz_distances_from_gram_dev(byz_ctx* ctx, const double* gram_dev, int64_t n_rows,
                                float* dist_dev, void* stream);
/* Near-duplicate pairs.  c_ii + c_jj - 2 c_ij cannot resolve rows that nearly coincide, the reference's  */
/* norm of the difference (defences.py:20) can.  byz_pairwise_distances_dev (and krum / bulyan) therefore  */
/* re-evaluate every pair with d^2 < (c_ii + c_jj)/16 on the difference itself.  A caller that only holds   */
/* a column slice of G (byz_distances_from_gram_dev on an all-reduced Gram) finishes the same step itself:  */
/* count -> per-rank sums of squared differences over the local columns -> (all-reduce) -> apply.           */
/* The list is ordered (ascending i, then j): slot p is the same pair on every GPU that holds the same Gram. */
/* A count > 0 OBLIGES the caller to finish with byz_near_pairs_apply_dev: until then listed entries hold    */
/* the Gram identity's value (possibly 0 by cancellation) and identical rows are not yet canonicalised.      */
int byz_near_pairs_count(byz_ctx* ctx, int64_t* count_host, void* stream);
int byz_near_pairs_sqdist_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                              const int32_t* row_index_dev, double* sq_dev, void* stream);
int byz_near_pairs_apply_dev(byz_ctx* ctx, const double* sq_dev, int64_t n_rows, float* dist_dev,
                             void* stream);

/* ---- defences.krum (reference defences.py:23-42) --------------------------------------- */
/* Selection loop only: ascending sort of every row's distances, sequential fp32 sum of the */
/* first (users_count - corrupted_count), candidates visited in order 1,0,2,3,... with a    */
/* strict '<' against 1e20.  *index_host = winner or -1.  scores_dev (optional, n_rows).    */
int byz_krum_select_dev(byz_ctx* ctx, const float* dist_dev, int64_t n_rows, int64_t users_count,
                        int64_t corrupted_count, int32_t* index_host, float* scores_dev,
                        void* stream);
/* Whole function.  check_assert != 0 applies `users_count >= 2*corrupted_count + 1`        */
/* (the reference skips it when return_index=True).  out_row_dev (optional) receives a copy */
/* of the winning row (index -1 selects the last row, as numpy's G[-1] does).               */
int byz_krum_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                 int64_t users_count, int64_t corrupted_count, int check_assert,
                 float* out_row_dev, int32_t* index_host, void* stream);

/* ---- defences.trimmed_mean (reference defences.py:44-52) ------------------------------- */
/* Per column: fp32 median, keep the k = n_rows - corrupted_count - 1 values closest to it  */
/* (ties in |x - med| by row order), out = mean(kept - med) + med.  row_index_dev (optional)*/
/* selects and orders the rows: row r of the logical matrix is G[row_index[r]].  k follows  */
/* Python slice semantics (k == 0 -> NaN, k < 0 -> drop from the far end).                  */
int byz_trimmed_mean_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols,
                         int64_t ld, const int32_t* row_index_dev, int64_t corrupted_count,
                         float* out_dev, void* stream);

/* 16-column tiles of the last byz_trimmed_mean_dev (or Bulyan second stage) that the fast ring selection */
/* handed to the general kernel (ties at the window edge, outliers, non-finite values); synchronises.    */
int byz_trimmed_mean_redone(byz_ctx* ctx, int64_t* tiles_host, void* stream);

/* ---- defences.bulyan (reference defences.py:55-70) ------------------------------------- */
/* Selection loop on a distance matrix: theta = users_count - 2*corrupted_count picks, each */
/* the Krum winner among the rows still present.  Exact fp64 running scores pick the winner  */
/* outright when it is clear; otherwise the contenders are re-scored with the reference's    */
/* sequential fp32 sums (defences.py:33-34), so the selection is the reference's own.        */
/* selection_dev: theta int32 indices in selection order.                                   */
int byz_bulyan_select_dev(byz_ctx* ctx, const float* dist_dev, int64_t n_rows, int64_t users_count,
                          int64_t corrupted_count, int32_t* selection_dev, void* stream);
/* Krum's index AND Bulyan's selection from one distance matrix with ONE sort of its rows (BASELINE       */
/* configs[4] runs both defences on the same distances): *krum_index_host as byz_krum_select_dev gives it, */
/* selection_dev as byz_bulyan_select_dev.                                                                  */
int byz_krum_bulyan_select_dev(byz_ctx* ctx, const float* dist_dev, int64_t n_rows, int64_t users_count,
                               int64_t corrupted_count, int32_t* krum_index_host, int32_t* selection_dev,
                               void* stream);
/* Rows the last selection loop had to re-score in the reference's sequential fp32 arithmetic because   */
/* their exact scores lay within that arithmetic's rounding band (0 for well separated clients).        */
int byz_bulyan_rescored(const byz_ctx* ctx, int64_t* rows_host);
/* Whole function (asserts users_count >= 4*corrupted_count + 3).  selection_dev optional.  */
int byz_bulyan_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                   int64_t users_count, int64_t corrupted_count, float* out_dev,
                   int32_t* selection_dev, void* stream);

/* ---- multi-GPU, columns layout: one context per GPU, the HOST owns the communicator ---- */
/* SURVEY.md 8(e)'s "cheaper equivalent": every rank holds ALL n_rows clients over its own slice of the      */
/* columns (G_local: n_rows x n_cols_local).  The path has ONE exchange: the n_rows x n_rows fp64 Gram of    */
/* the slices is summed over the ranks (plus, when clients nearly coincide, the short list of squared        */
/* differences of byz_near_pairs_*); the selection then runs replicated on every rank and the trimmed mean   */
/* on the local columns, so out_local is this rank's slice of the reference's result.  The library links no  */
/* collective library: the host passes its own in-place SUM all-reduce over the ranks, which must enqueue on */
/* `stream` and return 0 --                                                                                   */
/*     ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, (hipStream_t)stream)   (RCCL over xGMI)      */
/* with `comm` from ncclCommInitRank (one process per GPU) or ncclCommInitAll (one thread per GPU) reached     */
/* through `user`.  Every rank makes the same calls in the same order (the pair count is read from the same    */
/* all-reduced Gram on every rank).  A failing callback makes the call return BYZ_E_COLLECTIVE.  What          */
/* attacking_federate_learning_amd/sharded.py composes in Python over torch.distributed, for hosts without it. */
/* no_defense, trimmed_mean and the drift attack are independent per column: call the single-GPU entry         */
/* points on the local slice.                                                                                  */
typedef int (*byz_allreduce_f64_fn)(void* user, double* buf_dev, int64_t count, void* stream);
int byz_pairwise_distances_sharded_dev(byz_ctx* ctx, const float* G_local_dev, int64_t n_rows,
                                       int64_t n_cols_local, int64_t ld, byz_allreduce_f64_fn allreduce,
                                       void* user, float* dist_dev, void* stream);
/* defences.krum over the slices: *index_host is the reference's index (the same on every rank),              */
/* out_row_local_dev (optional) this rank's n_cols_local columns of the winning row.                           */
int byz_krum_sharded_dev(byz_ctx* ctx, const float* G_local_dev, int64_t n_rows, int64_t n_cols_local,
                         int64_t ld, int64_t users_count, int64_t corrupted_count, int check_assert,
                         byz_allreduce_f64_fn allreduce, void* user, float* out_row_local_dev,
                         int32_t* index_host, void* stream);
/* defences.bulyan over the slices: out_local_dev = this rank's n_cols_local columns of the aggregate,         */
/* selection_dev (optional) the theta selected clients in selection order (the same on every rank).            */
int byz_bulyan_sharded_dev(byz_ctx* ctx, const float* G_local_dev, int64_t n_rows, int64_t n_cols_local,
                           int64_t ld, int64_t users_count, int64_t corrupted_count,
                           byz_allreduce_f64_fn allreduce, void* user, float* out_local_dev,
                           int32_t* selection_dev, void* stream);

/* ---- malicious.Attack.attack / DriftAttack._attack_grads (malicious.py:10-36) ---------- */
/* Column mean and population std over the n_rows rows of G (the malicious clients' honest  */
/* gradients), drifted vector = mean - num_std * std.  Any output pointer may be NULL.      */
/* write_back != 0 also overwrites every row of G with the drifted vector (what             */
/* collect_gradients would copy in, server.py:81-83).                                       */
int byz_drift_attack_dev(byz_ctx* ctx, float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                         float num_std, float* drift_dev, float* mean_dev, float* std_dev,
                         int write_back, void* stream);
/* The same statistics when the rows are spread over several owners (the clients layout of   */
/* sharded.py: the malicious clients' rows sit on the first ranks).  numpy adds in row order,  */
/* so the additions form ONE chain through all owners: each calls byz_column_chain_dev on its  */
/* rows with the previous owner's output as carry_in (NULL for the first) -- out_sum[c] =      */
/* carry_in[c] + its rows' values in row order, or with `mean` their fl(fl(x - mean)^2) -- and */
/* the last owner ends a chain with byz_column_finish_dev: sum != NULL: mean = sum / total_rows */
/* (written); sumsq != NULL: std = sqrt(sumsq / total_rows), drift = mean - num_std * std (mean */
/* read where sum is NULL).  Bit for bit what byz_drift_attack_dev returns on the stacked rows. */
int byz_column_chain_dev(byz_ctx* ctx, const float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                         const float* carry_in_dev, const float* mean_dev, float* out_sum_dev, void* stream);
int byz_column_finish_dev(byz_ctx* ctx, const float* sum_dev, const float* sumsq_dev, int64_t total_rows,
                          float num_std, int64_t n_cols, float* mean_dev, float* std_dev, float* drift_dev,
                          void* stream);
/* The hook alone (malicious.py:34-36): mean[:] -= num_std * std[:], in place on the device. */
int byz_drift_axpy_dev(byz_ctx* ctx, float* mean_dev, const float* std_dev, int64_t n,
                       float num_std, void* stream);

/* ---- next row after the path: Server.defend's update (server.py:89-90) ----------------- */
/* velocity = momentum*velocity - lr*agg ; weights += velocity, fused, in place.            */
int byz_server_update_dev(byz_ctx* ctx, float* weights_dev, float* velocity_dev,
                          const float* agg_dev, int64_t n, float momentum, float learning_rate,
                          void* stream);

/* ---- next: BackdoorAttack._attack_grads without its training loop (backdoor.py:52-65) -- */
/* out = original_params - lr * grads_mean: the parameters the malicious network starts     */
/* from (backdoor.py:54).  fp32, numpy's operation order.                                   */
int byz_backdoor_initial_params_dev(byz_ctx* ctx, const float* original_params_dev,
                                    const float* grads_mean_dev, int64_t n, float learning_rate,
                                    float* out_dev, void* stream);
/* new_grads = ((params - lr*mean) - (mal_net_params + lr*mean)) / lr, clipped to            */
/* mean +- num_std * std (backdoor.py:57-63); np.clip's NaN behaviour.  Bit-identical to    */
/* numpy fp32 on the same inputs.                                                           */
int byz_backdoor_clip_dev(byz_ctx* ctx, const float* grads_mean_dev, const float* grads_stdev_dev,
                          const float* original_params_dev, const float* mal_net_params_dev,
                          int64_t n, float learning_rate, float num_std, float* out_dev,
                          void* stream);

/* ---- next: gradient assembly (user.py:92 np.concatenate + server.py:81-83 row copy) ----- */
/* Row `row` of the device-resident G = the concatenation of n_segments device tensors      */
/* (segments_dev: HOST array of device pointers, lengths: HOST array of element counts,     */
/* summing to n_cols), one launch per 32 tensors.  The _host form takes the client's        */
/* already-concatenated numpy vector (what usr.grads is in the reference).                  */
int byz_assemble_row_dev(byz_ctx* ctx, float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                         int64_t row, int64_t n_segments, const float* const* segments_dev,
                         const int64_t* lengths, void* stream);
/* Many clients in ONE launch, each with its own tensors (server.py:81-83's loop over users):  */
/* rows first_row .. first_row + n_clients - 1; segments_dev is a HOST array of n_clients x      */
/* n_segments device pointers, client-major (client c's tensor s at [c * n_segments + s]);       */
/* lengths (HOST, n_segments, summing to n_cols) is shared by all clients -- one model.  The     */
/* pointer table is copied to a context-owned device buffer on `stream`.                         */
int byz_assemble_rows_dev(byz_ctx* ctx, float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                          int64_t first_row, int64_t n_clients, int64_t n_segments,
                          const float* const* segments_dev, const int64_t* lengths, void* stream);
/* The same launch again with the pointer table of the LAST byz_assemble_rows_dev call on this */
/* context (it stays on the device): for rounds in which no client's gradient tensor moved,    */
/* which is the normal case -- a model's .grad buffers keep their addresses.  The caller       */
/* vouches that the tensors are the ones of that call; n_clients and n_segments must match it  */
/* (BYZ_E_INVALID otherwise, or when there has been no such call).  No host-to-device copy.    */
int byz_assemble_rows_again_dev(byz_ctx* ctx, float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                                int64_t first_row, int64_t n_clients, int64_t n_segments, void* stream);
/* Every client at once (what a batched client step produces): segment s is the row-major    */
/* (n_rows x lengths[s]) gradient of parameter s for all clients; G[:, start_s:start_s+len_s] */
/* := that block.  One launch per 32 parameters.                                             */
int byz_assemble_columns_dev(byz_ctx* ctx, float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                             int64_t n_segments, const float* const* segments_dev,
                             const int64_t* lengths, void* stream);
int byz_assemble_row_host(byz_ctx* ctx, float* G_dev, int64_t n_rows, int64_t n_cols, int64_t ld,
                          int64_t row, const float* grads_host, void* stream);

/* ---- host-pointer convenience (what the numpy drop-in uses) ---------------------------- */
/* G_host is the reference's C-contiguous np.float32 users_grads.  name: 0 NoDefense, 1 Krum, */
/* 2 TrimmedMean, 3 Bulyan (the keys of defences.defend, defences.py:73-75).  out_host: n_cols */
/* floats.  aux_host (optional): Krum -> 1 int32 (index), Bulyan -> theta int32 (selection).  */
int byz_defend_host(byz_ctx* ctx, int name, const float* G_host, int64_t n_rows, int64_t n_cols,
                    int64_t users_count, int64_t corrupted_count, int check_assert,
                    float* out_host, int32_t* aux_host);
int byz_pairwise_distances_host(byz_ctx* ctx, const float* G_host, int64_t n_rows, int64_t n_cols,
                                float* dist_host);
int byz_krum_select_host(byz_ctx* ctx, const float* dist_host, int64_t n_rows, int64_t users_count,
                         int64_t corrupted_count, int32_t* index_host);
int byz_drift_attack_host(byz_ctx* ctx, const float* rows_host, int64_t n_rows, int64_t n_cols,
                          float num_std, float* drift_host, float* mean_host, float* std_host);

/* ---- per-kernel timing (bench.py's roofline leg) --------------------------------------- */
/* When enabled, every kernel launch is bracketed by HIP events on its own stream.          */
enum {
    BYZ_K_COLUMN_STATS = 0, BYZ_K_GRAM = 1, BYZ_K_GRAM_REDUCE = 2, BYZ_K_DISTANCES = 3,
    BYZ_K_ROW_SORT = 4, BYZ_K_KRUM_ARGMIN = 5, BYZ_K_BULYAN_LOOP = 6, BYZ_K_TRIMMED_MEAN = 7,
    BYZ_K_MISC = 8, BYZ_K_PLANE_SPLIT = 9, BYZ_K_COUNT = 10
};
int byz_timing_enable(byz_ctx* ctx, int on);
int byz_timing_reset(byz_ctx* ctx);
int byz_timing_read(byz_ctx* ctx, int kernel, double* total_ms, int64_t* launches);
const char* byz_kernel_name(int kernel);

/* ---- self-test of the cross-lane exchange primitives (tests only) ----------------------- */
/* out_dev: 64 * n_patterns int32; entry [p*64 + lane] = source lane observed by `lane`.     */
int byz_selftest_lane_exchange_dev(byz_ctx* ctx, int32_t* out_dev, int32_t* n_patterns_host,
                                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BYZAGG_H */
