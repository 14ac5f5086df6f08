/*
 * CPU oracle for the capacity-aware beam search -- TEST INFRASTRUCTURE, never the product path.
 *
 * Plain C restatement of src/gcn/algorithms/beamsearch_uelb.py of the reference (file:line
 * citations below are relative to the reference checkout).  Built by oracle/Makefile into
 * oracle/_build/libbeam_oracle.so and loaded through ctypes by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg only.  It is pinned against vectors the reference itself
 * produced (oracle/gen_golden.py -> tests/golden/beam_*.npz).
 *
 * Score arithmetic is fp32, added left to right along the path, as the reference's 0-d float
 * tensors do (beamsearch_uelb.py:158-159).  Ordering is "score descending, insertion order
 * ascending", which is what Python's stable sorted(..., reverse=True) gives (:144, :173).
 */
#include <stdlib.h>
#include <string.h>

typedef struct {
    float score;
    int order;   /* insertion index, breaks ties */
    int node;
    int len;
    int *path;   /* len entries */
} Item;

static int cmp_items(const void *a, const void *b) {
    const Item *x = (const Item *)a, *y = (const Item *)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return (x->order > y->order) - (x->order < y->order);
}

/* _beam_search_for_commodity (:137-177) / _unconstrained_beam_search_for_commodity (:417-457).
 * cap: [V,V] capacity at the start of this commodity; y: log-probabilities with element
 * (i, j) at y[(i*V + j)*ystride].  Writes the best path to out, returns its length (0 = none). */
static int search_commodity(const float *cap, const float *y, long ystride, int V, int src, int dst,
                            float demand, int K, int check_demand, int *out) {
    int maxq = K * V + 1;
    Item *q = (Item *)malloc(sizeof(Item) * (size_t)maxq);
    Item *nq = (Item *)malloc(sizeof(Item) * (size_t)maxq);
    int *pool_a = (int *)malloc(sizeof(int) * (size_t)maxq * (size_t)V);
    int *pool_b = (int *)malloc(sizeof(int) * (size_t)maxq * (size_t)V);
    int nqn, qn = 1, have_best = 0, best_len = 0, v, i, n;
    float best_score = 0.0f;
    q[0].score = 0.0f; q[0].order = 0; q[0].node = src; q[0].len = 1; q[0].path = pool_a; pool_a[0] = src;
    while (qn > 0) {
        qsort(q, (size_t)qn, sizeof(Item), cmp_items);
        if (qn > K) qn = K;
        nqn = 0;
        for (i = 0; i < qn; ++i) {
            const Item *it = &q[i];
            if (it->node == dst) {
                /* best_paths keeps discovery order and is stably sorted at the end (:173), so
                 * the first path with the strictly highest score wins */
                if (!have_best || it->score > best_score) {
                    have_best = 1; best_score = it->score; best_len = it->len;
                    memcpy(out, it->path, sizeof(int) * (size_t)it->len);
                }
                continue;
            }
            for (n = 0; n < V; ++n) {
                int seen = 0;
                float c = cap[it->node * V + n];
                for (v = 0; v < it->len; ++v) if (it->path[v] == n) { seen = 1; break; }
                if (seen) continue;                       /* :153 */
                if (c == 0.0f) continue;                  /* :155 */
                if (check_demand && !(demand <= c)) continue; /* :161 (remaining == entry capacity on a simple path) */
                {
                    Item *d = &nq[nqn];
                    d->score = it->score + y[((long)it->node * V + n) * ystride];
                    d->order = nqn; d->node = n; d->len = it->len + 1;
                    d->path = pool_b + (size_t)nqn * (size_t)V;
                    memcpy(d->path, it->path, sizeof(int) * (size_t)it->len);
                    d->path[it->len] = n;
                    ++nqn;
                }
            }
        }
        { Item *t = q; q = nq; nq = t; }
        { int *t = pool_a; pool_a = pool_b; pool_b = t; }
        qn = nqn;
    }
    free(q); free(nq); free(pool_a); free(pool_b);
    return have_best ? best_len : 0;
}

/* algorithm: 0 standard, 1 deterministic, 2 greedy, 3 unconstrained (BeamSearchFactory, :463-468).
 * y [B,V,V,C] f32, cap [B,V,V] f32, comm [B,C,3] int64, perms [n_perms,C] int32 consumed one per
 * attempt graph after graph (:99, :299).  paths [B,C,PL] int32 padded with -1 where PL = max(V,10),
 * path_len [B,C], feasible [B], attempts [B].
 * Returns the number of permutations consumed, or -1 if perms ran out. */
int beam_oracle_search(const float *y, const float *cap, const long long *comm, int B, int V, int C,
                       int K, int algorithm, int strict, int max_iter, const int *perms, int n_perms,
                       int *paths, int *path_len, int *feasible, int *attempts) {
    int PL = V > 10 ? V : 10;
    int used = 0, b, c, pos, k;
    float *cur = (float *)malloc(sizeof(float) * (size_t)V * (size_t)V);
    int *found = (int *)malloc(sizeof(int) * (size_t)C * (size_t)V);
    int *found_len = (int *)malloc(sizeof(int) * (size_t)C);
    int *order = (int *)malloc(sizeof(int) * (size_t)C);
    if (algorithm == 2) K = 1;                           /* GreedyBeamSearch.__init__, :286-287 */
    for (b = 0; b < B; ++b) {
        const float *yb = y + (size_t)b * V * V * C;
        const float *capb = cap + (size_t)b * V * V;
        const long long *cm = comm + (size_t)b * C * 3;
        int *pb = paths + (size_t)b * C * PL;
        int *plb = path_len + (size_t)b * C;
        for (k = 0; k < C * PL; ++k) pb[k] = -1;
        attempts[b] = 0;
        if (algorithm == 3) {                            /* UnconstrainedBeamSearch, :387-415 */
            for (c = 0; c < C; ++c)
                plb[c] = search_commodity(capb, yb + c, C, V, (int)cm[c * 3], (int)cm[c * 3 + 1], 0.0f, K, 0,
                                          pb + (size_t)c * PL);
            feasible[b] = 1;
            continue;
        }
        {
            int count = 0;
            while (count < max_iter) {
                int nfound = 0;
                if (algorithm == 1) { for (c = 0; c < C; ++c) order[c] = c; }
                else {
                    if (used >= n_perms) { free(cur); free(found); free(found_len); free(order); return -1; }
                    memcpy(order, perms + (size_t)used * C, sizeof(int) * (size_t)C);
                    ++used;
                }
                ++attempts[b];
                memcpy(cur, capb, sizeof(float) * (size_t)V * (size_t)V);
                for (pos = 0; pos < C; ++pos) {
                    int ci = order[pos];
                    float d = (float)cm[ci * 3 + 2];
                    int len = search_commodity(cur, yb + ci, C, V, (int)cm[ci * 3], (int)cm[ci * 3 + 1], d, K, 1,
                                               found + (size_t)pos * V);
                    if (len == 0) break;                 /* :119-120 */
                    found_len[pos] = len;
                    if (strict)                          /* :123-124 */
                        for (k = 0; k + 1 < len; ++k) {
                            int u = found[(size_t)pos * V + k], w = found[(size_t)pos * V + k + 1];
                            cur[u * V + w] -= d;
                        }
                    ++nfound;
                }
                for (k = 0; k < C * PL; ++k) pb[k] = -1;
                if (nfound == C) {
                    for (pos = 0; pos < C; ++pos) {      /* un-permute, :102,129 */
                        int ci = order[pos];
                        memcpy(pb + (size_t)ci * PL, found + (size_t)pos * V, sizeof(int) * (size_t)found_len[pos]);
                        plb[ci] = found_len[pos];
                    }
                    feasible[b] = 1;
                    count = max_iter;
                } else {                                 /* fallback, :131-133,179-181 */
                    for (c = 0; c < C; ++c) {
                        for (k = 0; k < 10; ++k) pb[(size_t)c * PL + k] = k;
                        plb[c] = 10;
                    }
                    feasible[b] = 0;
                    ++count;
                }
            }
        }
    }
    free(cur); free(found); free(found_len); free(order);
    return used;
}
