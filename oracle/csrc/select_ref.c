/*
 * CPU restatement of the reference's Krum / Bulyan selection at sizes the Python oracle cannot finish.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing in the product links or loads this.
 *
 * What is restated (reference = shaneson0/attacking_federate_learning):
 *   defences.py:26-37   krum: for every user still in the dict, in dict order 1, 0, 2, 3, ...:
 *                         errors = sorted(distances[user].values())
 *                         current_error = sum(errors[:users_count - corrupted_count])
 *                         strict '<' against a running minimum that starts at 1e20 / index -1
 *   defences.py:59-68   bulyan: theta = users_count - 2*corrupted_count picks, each
 *                         krum(..., users_count - len(selection_set), corrupted_count, distances, True),
 *                         then the winner's row and column leave the dict.
 * Arithmetic: `sum` over np.float32 scalars under numpy >= 2 is a left-to-right fp32 sum (mode 0, what
 * oracle/faithful.py pins bit-for-bit against the reference); mode 1 evaluates the same rule with fp64
 * sums (oracle/ideal.py) and reports the relative margin of every pick.
 *
 * The dict of dicts is a dense n x n fp32 matrix here (diagonal ignored).  Every row is sorted ONCE
 * (ascending value; equal values in any order: a sum does not see it); a pick walks each live row's sorted
 * list, skipping removed columns, exactly the list the reference's sorted() would produce at that moment.
 * O(theta * n^2) work, rows in parallel with OpenMP.  Finite, non-NaN distances only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    float v;
    int32_t c;
} entry_t;

static int cmp_entry(const void* a, const void* b) {
    const entry_t* x = (const entry_t*)a;
    const entry_t* y = (const entry_t*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->c > y->c) - (x->c < y->c);
}

/* number of items some_list[:stop] keeps (Python slice semantics, defences.py:34) */
static int64_t prefix_len(int64_t length, int64_t stop) {
    if (stop >= 0) return stop < length ? stop : length;
    return length + stop > 0 ? length + stop : 0;
}

static int visit_row(int k, int n) { /* k-th key of the dict: 1, 0, 2, 3, ... (defences.py:17-20) */
    (void)n;
    return k == 0 ? 1 : (k == 1 ? 0 : k);
}

typedef struct {
    int n;
    entry_t* sorted; /* n rows of n-1 entries */
} table_t;

static int build_table(const float* dist, int n, table_t* t) {
    t->n = n;
    t->sorted = (entry_t*)malloc((size_t)n * (size_t)(n > 1 ? n - 1 : 1) * sizeof(entry_t));
    if (!t->sorted) return -1;
#pragma omp parallel for schedule(static)
    for (int u = 0; u < n; ++u) {
        entry_t* row = t->sorted + (size_t)u * (n - 1);
        int k = 0;
        for (int c = 0; c < n; ++c)
            if (c != u) {
                row[k].v = dist[(size_t)u * n + c];
                row[k].c = c;
                ++k;
            }
        qsort(row, (size_t)(n - 1), sizeof(entry_t), cmp_entry);
    }
    return 0;
}

/* score of live row u with `live` rows present: the first `keep` live entries of its sorted list */
static double row_score(const table_t* t, int u, const uint8_t* removed, int64_t take, int mode) {
    const entry_t* row = t->sorted + (size_t)u * (t->n - 1);
    int64_t got = 0;
    if (mode == 0) {
        float s = 0.0f; /* Python's int 0 + np.float32 is exact */
        for (int r = 0; r < t->n - 1 && got < take; ++r)
            if (!removed[row[r].c]) {
                s = s + row[r].v;
                ++got;
            }
        return (double)s;
    }
    double s = 0.0;
    for (int r = 0; r < t->n - 1 && got < take; ++r)
        if (!removed[row[r].c]) {
            s += (double)row[r].v;
            ++got;
        }
    return s;
}

/* one krum(..., return_index=True) call on the live rows; scores_out (n doubles, optional) gets every
 * live row's score (removed rows: +inf); margin_out (optional) the relative gap to the runner-up */
static int pick(const table_t* t, const uint8_t* removed, int live, int64_t users_count, int64_t corrupted,
                int mode, double* scores, double* margin_out) {
    const int n = t->n;
    const int64_t take = prefix_len((int64_t)live - 1, users_count - corrupted);
#pragma omp parallel for schedule(dynamic, 16)
    for (int u = 0; u < n; ++u) scores[u] = removed[u] ? INFINITY : row_score(t, u, removed, take, mode);
    double best = 1e20;
    int best_idx = -1;
    if (n >= 2) { /* a single row has an empty dict: nothing is visited */
        for (int k = 0; k < n; ++k) {
            const int u = visit_row(k, n);
            if (removed[u]) continue;
            if (scores[u] < best) {
                best = scores[u];
                best_idx = u;
            }
        }
    }
    if (margin_out) {
        double second = INFINITY;
        for (int u = 0; u < n; ++u)
            if (!removed[u] && u != best_idx && scores[u] < second) second = scores[u];
        const double denom = fabs(best) > 1e-300 ? fabs(best) : 1e-300;
        *margin_out = (best_idx >= 0 && isfinite(second)) ? (second - best) / denom : INFINITY;
    }
    return best_idx;
}

int ref_set_threads(int n_threads) {
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
    return omp_get_max_threads();
#else
    (void)n_threads;
    return 1;
#endif
}

/* defences.py:23-42 with return_index=True.  Returns the index (or -1); -2 on allocation failure. */
int ref_krum_pick(const float* dist, int n, int64_t users_count, int64_t corrupted, int mode, double* scores_out,
                  double* margin_out) {
    table_t t;
    if (build_table(dist, n, &t)) return -2;
    uint8_t* removed = (uint8_t*)calloc((size_t)n, 1);
    double* scores = scores_out ? scores_out : (double*)malloc((size_t)n * sizeof(double));
    const int idx = pick(&t, removed, n, users_count, corrupted, mode, scores, margin_out);
    if (!scores_out) free(scores);
    free(removed);
    free(t.sorted);
    return idx;
}

/* defences.py:59-68.  selection: theta int32 (selection order); margins: theta doubles or NULL.
 * Returns the number of picks made: theta, or fewer when a pick found no score below 1e20 (the reference
 * raises KeyError(-1) there); -2 on allocation failure. */
int ref_bulyan_selection(const float* dist, int n, int64_t users_count, int64_t corrupted, int mode,
                         int32_t* selection, double* margins) {
    const int64_t theta = users_count - 2 * corrupted;
    table_t t;
    if (build_table(dist, n, &t)) return -2;
    uint8_t* removed = (uint8_t*)calloc((size_t)n, 1);
    double* scores = (double*)malloc((size_t)n * sizeof(double));
    int made = 0;
    for (int64_t k = 0; k < theta; ++k) {
        const int idx = pick(&t, removed, n - (int)k, users_count - k, corrupted, mode, scores,
                             margins ? margins + k : NULL);
        if (idx < 0) break;
        selection[k] = idx;
        removed[idx] = 1;
        ++made;
    }
    free(scores);
    free(removed);
    free(t.sorted);
    return made;
}

/* Spot check of a selection too long to recompute: for every listed pick t the rows selection[0..t-1] are removed
 * and the reference's pick among the rest (defences.py:26-37, one full scoring pass) must be selection[t].
 * Returns the number of listed picks that disagree; first_bad gets the first such pick (or -1).
 * If every pick of a selection passes, the selection IS the reference's (induction over t); a sample of picks at a
 * size where all of them cannot be afforded is evidence, not proof, and the tests say which one they ran. */
int ref_verify_picks(const float* dist, int n, int64_t users_count, int64_t corrupted, int mode, const int32_t* selection,
                     int theta, const int32_t* picks, int n_picks, int32_t* first_bad, int32_t* expected) {
    table_t t;
    if (build_table(dist, n, &t)) return -2;
    uint8_t* removed = (uint8_t*)calloc((size_t)n, 1);
    double* scores = (double*)malloc((size_t)n * sizeof(double));
    int bad = 0;
    *first_bad = -1;
    *expected = -1;
    for (int k = 0; k < n_picks; ++k) {
        const int at = picks[k];
        if (at < 0 || at >= theta) continue;
        memset(removed, 0, (size_t)n);
        for (int q = 0; q < at; ++q) removed[selection[q]] = 1;
        const int idx = pick(&t, removed, n - at, users_count - at, corrupted, mode, scores, NULL);
        if (idx != selection[at]) {
            if (bad == 0) {
                *first_bad = at;
                *expected = idx;
            }
            ++bad;
        }
    }
    free(scores);
    free(removed);
    free(t.sorted);
    return bad;
}

/* The margin protocol's second clause: a selection that was made on DIFFERENT numbers (the engine's
 * distances, fp32 sums) is replayed on these: at every pick t the rows selection[0..t-1] are removed -- the state is the
 * selection's OWN, not the oracle's -- every live row is scored (defences.py:26-37, `mode` arithmetic), and
 *   excess[t] = score(selection[t]) / min score - 1      how far from the optimum the selection's pick is (0: it IS an argmin)
 *   margin[t] = (runner-up - min) / min                  how contested the pick was
 *   argmin[t] = the row the rule itself picks in that state (visit order 1, 0, 2, ..., strict '<')
 * A pick is "inside tau" when margin[t] <= tau; the protocol bounds excess[t] for every pick and counts the contested ones.
 * O(theta n^2), rows in parallel.  Returns theta, or -2 on allocation failure, -3 when the selection repeats a row. */
int ref_replay_selection(const float* dist, int n, int64_t users_count, int64_t corrupted, int mode, const int32_t* selection,
                         int theta, double* excess, double* margin, int32_t* argmin) {
    table_t t;
    if (build_table(dist, n, &t)) return -2;
    uint8_t* removed = (uint8_t*)calloc((size_t)n, 1);
    double* scores = (double*)malloc((size_t)n * sizeof(double));
    int rc = theta;
    for (int k = 0; k < theta; ++k) {
        const int mine = selection[k];
        if (mine < 0 || mine >= n || removed[mine]) {
            rc = -3;
            break;
        }
        double m = 0.0;
        const int idx = pick(&t, removed, n - k, users_count - k, corrupted, mode, scores, &m);
        double best = INFINITY;
        for (int u = 0; u < n; ++u)
            if (!removed[u] && scores[u] < best) best = scores[u];
        const double denom = fabs(best) > 1e-300 ? fabs(best) : 1e-300;
        excess[k] = (scores[mine] - best) / denom;
        margin[k] = m;
        argmin[k] = idx;
        removed[mine] = 1;
    }
    free(scores);
    free(removed);
    free(t.sorted);
    return rc;
}
