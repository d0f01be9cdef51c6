/*
 * A C host that shards Bulyan (reference defences.py:55-70) and Krum (defences.py:23-42) over the GPUs of one node through
 * the C ABI alone: one thread per GPU, the columns layout (every GPU holds all N clients over its slice of the columns;
 * SURVEY.md 8(e)), RCCL's ncclAllReduce behind the library's all-reduce callback (include/byzagg.h, "multi-GPU, columns
 * layout").  No Python, no torch.
 *
 *   gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/shard_columns.c -o shard_columns \
 *       -L attacking_federate_learning_amd -lbyzagg -L /opt/rocm/lib -lrccl -lamdhip64 -lpthread -lm \
 *       -Wl,-rpath,$PWD/attacking_federate_learning_amd -Wl,-rpath,/opt/rocm/lib
 *   ./shard_columns <gpus> <clients N> <params D> <corrupted f>
 *
 * The matrix is synthetic (a fixed generator, so every run sees the same clients): rows 0 .. f-1 are ONE vector, as the
 * reference's attack leaves them (malicious.py:26-27).  Every GPU must arrive at the same selection; GPU 0 then repeats
 * the round UNSHARDED on the whole matrix (byz_bulyan_dev / byz_krum_dev) and the program checks that the sharded
 * selection and Krum index are the unsharded ones and that the slices of the aggregate agree to 1e-5 (north_star's
 * tolerance).  Exit status 0 and "OK" on success.  tests/test_gpu_sharded_cabi.py builds and runs it with one GPU.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "byzagg.h"

#define MAX_GPUS 16

typedef struct {
    int rank, world;
    int64_t n, d, f, lo, hi;          /* this rank's columns [lo, hi) */
    const float* g_host;              /* the whole matrix (row-major n x d), shared by the threads */
    ncclComm_t comm;
    byz_ctx* ctx;
    float* out_host;                  /* this rank's slice of the aggregate */
    int32_t* selection_host;          /* theta indices */
    int32_t krum_index;
    int status;
    char error[256];
} rank_t;

/* byz_allreduce_f64_fn: the host's in-place sum over the ranks, on the call's stream */
static int allreduce_over_rccl(void* user, double* buf, int64_t count, void* stream) {
    rank_t* r = (rank_t*)user;
    return ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, r->comm, (hipStream_t)stream) == ncclSuccess ? 0 : 1;
}

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != BYZ_OK) {                                                                         \
            snprintf(r->error, sizeof r->error, "%s -> %d: %s", #call, rc_, byz_last_error());       \
            r->status = rc_;                                                                         \
            return NULL;                                                                             \
        }                                                                                            \
    } while (0)

static void* run_rank(void* arg) {
    rank_t* r = (rank_t*)arg;
    const int64_t n = r->n, w = r->hi - r->lo, theta = r->n - 2 * r->f;
    void *g_dev = NULL, *out_dev = NULL, *sel_dev = NULL;
    CHECK(byz_ctx_create(r->rank, &r->ctx));            /* GPU `rank`; the library makes it current on this thread per call */
    CHECK(byz_malloc(r->ctx, n * w * (int64_t)sizeof(float), &g_dev));
    CHECK(byz_malloc(r->ctx, w * (int64_t)sizeof(float), &out_dev));
    CHECK(byz_malloc(r->ctx, theta * (int64_t)sizeof(int32_t), &sel_dev));
    /* this rank's columns of Server.users_grads (server.py:81-83): a strided copy out of the host matrix */
    CHECK(byz_upload_2d(r->ctx, g_dev, w * (int64_t)sizeof(float), r->g_host + r->lo, r->d * (int64_t)sizeof(float),
                        w * (int64_t)sizeof(float), n, NULL));
    CHECK(byz_stream_sync(r->ctx, NULL));
    CHECK(byz_krum_sharded_dev(r->ctx, (const float*)g_dev, n, w, w, n, r->f, 1, allreduce_over_rccl, r, NULL, &r->krum_index, NULL));
    CHECK(byz_bulyan_sharded_dev(r->ctx, (const float*)g_dev, n, w, w, n, r->f, allreduce_over_rccl, r, (float*)out_dev,
                                 (int32_t*)sel_dev, NULL));
    CHECK(byz_ctx_check(r->ctx, NULL));
    CHECK(byz_download(r->ctx, r->out_host, out_dev, w * (int64_t)sizeof(float), NULL));
    CHECK(byz_download(r->ctx, r->selection_host, sel_dev, theta * (int64_t)sizeof(int32_t), NULL));
    byz_free(r->ctx, g_dev);
    byz_free(r->ctx, out_dev);
    byz_free(r->ctx, sel_dev);
    return NULL;
}

/* the whole round on ONE GPU, for the comparison */
static int run_unsharded(rank_t* r, float* out_host, int32_t* selection_host, int32_t* krum_index) {
    const int64_t n = r->n, d = r->d, theta = n - 2 * r->f;
    void *g_dev = NULL, *out_dev = NULL, *sel_dev = NULL;
#define CHECK1(call)                                                                                 \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != BYZ_OK) {                                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, byz_last_error());                         \
            return rc_;                                                                              \
        }                                                                                            \
    } while (0)
    CHECK1(byz_malloc(r->ctx, n * d * (int64_t)sizeof(float), &g_dev));
    CHECK1(byz_malloc(r->ctx, d * (int64_t)sizeof(float), &out_dev));
    CHECK1(byz_malloc(r->ctx, theta * (int64_t)sizeof(int32_t), &sel_dev));
    CHECK1(byz_upload(r->ctx, g_dev, r->g_host, n * d * (int64_t)sizeof(float), NULL));
    CHECK1(byz_stream_sync(r->ctx, NULL));
    CHECK1(byz_krum_dev(r->ctx, (const float*)g_dev, n, d, d, n, r->f, 1, NULL, krum_index, NULL));
    CHECK1(byz_bulyan_dev(r->ctx, (const float*)g_dev, n, d, d, n, r->f, (float*)out_dev, (int32_t*)sel_dev, NULL));
    CHECK1(byz_download(r->ctx, out_host, out_dev, d * (int64_t)sizeof(float), NULL));
    CHECK1(byz_download(r->ctx, selection_host, sel_dev, theta * (int64_t)sizeof(int32_t), NULL));
    byz_free(r->ctx, g_dev);
    byz_free(r->ctx, out_dev);
    byz_free(r->ctx, sel_dev);
    return BYZ_OK;
#undef CHECK1
}

/* a small fixed generator: uniform in (-1, 1), rows scaled by 1 .. 1.5 (the 'scaled' family of the tests) */
static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static float next_uniform(void) {
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((double)(lcg_state >> 40) / (double)(1ull << 23) - 1.0);
}

int main(int argc, char** argv) {
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    const int64_t n = argc > 2 ? atoll(argv[2]) : 200, d = argc > 3 ? atoll(argv[3]) : 5000, f = argc > 4 ? atoll(argv[4]) : 40;
    if (world < 1 || world > MAX_GPUS || n < 4 * f + 3 || d < world || f < 0) {
        fprintf(stderr, "usage: %s <gpus 1..%d> <clients N >= 4 f + 3> <params D >= gpus> <corrupted f>\n", argv[0], MAX_GPUS);
        return 2;
    }
    const int64_t theta = n - 2 * f;
    float* g = (float*)malloc((size_t)(n * d) * sizeof(float));
    for (int64_t i = 0; i < n; ++i) {
        const float scale = 1.0f + 0.5f * (float)((i * 7919) % n) / (float)n;
        for (int64_t k = 0; k < d; ++k) g[i * d + k] = scale * next_uniform();
    }
    /* the attack as the reference leaves it: the f malicious rows are one vector (mean - 1.5 std of their honest values) */
    for (int64_t k = 0; k < d && f > 0; ++k) {
        double mean = 0.0, var = 0.0;
        for (int64_t i = 0; i < f; ++i) mean += g[i * d + k];
        mean /= (double)f;
        for (int64_t i = 0; i < f; ++i) var += (g[i * d + k] - mean) * (g[i * d + k] - mean);
        const float drifted = (float)(mean - 1.5 * sqrt(var / (double)f));
        for (int64_t i = 0; i < f; ++i) g[i * d + k] = drifted;
    }

    int devices[MAX_GPUS];
    ncclComm_t comms[MAX_GPUS];
    for (int p = 0; p < world; ++p) devices[p] = p;
    if (ncclCommInitAll(comms, world, devices) != ncclSuccess) {
        fprintf(stderr, "ncclCommInitAll over %d GPU(s) failed\n", world);
        return 1;
    }
    rank_t ranks[MAX_GPUS];
    pthread_t threads[MAX_GPUS];
    memset(ranks, 0, sizeof ranks);
    for (int p = 0; p < world; ++p) {
        rank_t* r = &ranks[p];
        r->rank = p;
        r->world = world;
        r->n = n;
        r->d = d;
        r->f = f;
        r->lo = d * p / world;
        r->hi = d * (p + 1) / world;
        r->g_host = g;
        r->comm = comms[p];
        r->out_host = (float*)malloc((size_t)(r->hi - r->lo) * sizeof(float));
        r->selection_host = (int32_t*)malloc((size_t)theta * sizeof(int32_t));
        pthread_create(&threads[p], NULL, run_rank, r);
    }
    int failed = 0;
    for (int p = 0; p < world; ++p) {
        pthread_join(threads[p], NULL);
        if (ranks[p].status != BYZ_OK) {
            fprintf(stderr, "GPU %d: %s\n", p, ranks[p].error);
            failed = 1;
        }
    }
    if (failed) return 1;
    for (int p = 1; p < world; ++p) {
        if (memcmp(ranks[p].selection_host, ranks[0].selection_host, (size_t)theta * sizeof(int32_t)) != 0 ||
            ranks[p].krum_index != ranks[0].krum_index) {
            fprintf(stderr, "GPU %d disagrees with GPU 0 on the selection\n", p);
            return 1;
        }
    }
    /* the same round unsharded on GPU 0 */
    float* out_full = (float*)malloc((size_t)d * sizeof(float));
    int32_t* sel_full = (int32_t*)malloc((size_t)theta * sizeof(int32_t));
    int32_t krum_full = -2;
    if (run_unsharded(&ranks[0], out_full, sel_full, &krum_full) != BYZ_OK) return 1;
    if (krum_full != ranks[0].krum_index) {
        fprintf(stderr, "Krum: sharded index %d, unsharded %d\n", ranks[0].krum_index, krum_full);
        return 1;
    }
    int64_t differing = 0;
    for (int64_t t = 0; t < theta; ++t) differing += sel_full[t] != ranks[0].selection_host[t];
    if (differing != 0) {
        /* two scores apart by the last bits of a distance may swap two picks between two summation orders of the Gram (the
         * slices' fp64 partial Grams are added in another order than one GPU's chunks): reported; the SETS must still agree */
        fprintf(stderr, "Bulyan: %lld of %lld picks differ in ORDER between the sharded and the unsharded run\n", (long long)differing,
                (long long)theta);
        unsigned char* seen = (unsigned char*)calloc((size_t)n, 1);
        for (int64_t t = 0; t < theta; ++t) seen[sel_full[t]] = 1;
        for (int64_t t = 0; t < theta; ++t)
            if (!seen[ranks[0].selection_host[t]]) {
                fprintf(stderr, "Bulyan: client %d is selected by the sharded run only\n", ranks[0].selection_host[t]);
                return 1;
            }
        free(seen);
    }
    double worst = 0.0;
    for (int p = 0; p < world; ++p)
        for (int64_t k = ranks[p].lo; k < ranks[p].hi; ++k) {
            const double a = ranks[p].out_host[k - ranks[p].lo], b = out_full[k];
            const double err = fabs(a - b) / (1e-5 + 1e-5 * fabs(b));
            if (err > worst) worst = err;
        }
    printf("gpus %d  N %lld  D %lld  f %lld  theta %lld | krum index %d | bulyan picks %d %d %d ... | aggregate vs unsharded: "
           "worst error %.3g of the 1e-5 tolerance | picks in another order: %lld\n",
           world, (long long)n, (long long)d, (long long)f, (long long)theta, krum_full, ranks[0].selection_host[0],
           ranks[0].selection_host[1], ranks[0].selection_host[2], worst, (long long)differing);
    for (int p = 0; p < world; ++p) {
        byz_ctx_destroy(ranks[p].ctx);
        ncclCommDestroy(comms[p]);
    }
    if (worst > 1.0) {
        fprintf(stderr, "aggregate differs beyond 1e-5\n");
        return 1;
    }
    printf("OK\n");
    return 0;
}
