/*
 * oracle/ward_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp32, amd64 semantics: no FMA contraction) of the
 * reference's size-constrained Ward clustering,
 *     /root/reference/internal/clustering/clustering.go   (whole file).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product library never links it.
 *
 * Parity status: the reference ships no tests or golden vectors for this path
 *  and no Go toolchain exists here, so this restatement is pinned
 * by the hand-derived known-answer tests KAT-1..KAT-7 of SURVEY.md 8c
 * (tests/test_oracle_ward.py) -> "parity pinned by hand-derived KATs only".
 *
 * The restatement is deliberately LITERAL: position-compacted cluster list,
 * dense row-pointer distance matrix with order-preserving row/column deletion,
 * full lower-triangle scan per iteration.  It is O(N^3) like the reference.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ICL_REF_MAXF FLT_MAX /* math.MaxFloat32, clustering.go:120,230 */

/* clustering.go:148-157 DotFloat32: sequential fp32 sum, product rounded, then sum rounded. */
float icl_ref_dot(const float *a, const float *b, int64_t d)
{
    float sum = 0.0f;
    for (int64_t i = 0; i < d; ++i) {
        float p = a[i] * b[i];
        sum = sum + p;
    }
    return sum;
}

/* clustering.go:136-145 WardDistance: diff vector, dot(diff,diff), (float(sa*sb)/float(sa+sb))*dot. */
float icl_ref_ward_distance(const float *ca, int64_t sa, const float *cb, int64_t sb, int64_t d)
{
    float sum = 0.0f;
    for (int64_t i = 0; i < d; ++i) {
        float diff = ca[i] - cb[i];    /* :139 */
        float p = diff * diff;         /* :154 product */
        sum = sum + p;                 /* :154 accumulate */
    }
    float numerator = (float)(sa * sb);   /* :142 int product, then convert */
    float denominator = (float)(sa + sb); /* :143 */
    return (numerator / denominator) * sum; /* :144 */
}

/* clustering.go:37-40 centroid of MergeClusters(a,b). */
void icl_ref_merge_centroid(const float *ca, int64_t sa, const float *cb, int64_t sb, int64_t d, float *out)
{
    float fa = (float)sa, fb = (float)sb, fs = (float)(sa + sb);
    for (int64_t i = 0; i < d; ++i) {
        float pa = fa * ca[i];
        float pb = fb * cb[i];
        float s = pa + pb;
        out[i] = s / fs;
    }
}

/* clustering.go:61-73 ComputeInitialDistanceMatrix on flat row-major centroids C[n][d];
 * D is n x n with leading dimension ld; diagonal left 0 (never written by the reference). */
void icl_ref_initial_distance_matrix(const float *C, const int32_t *sizes, int64_t n, int64_t d,
                                     float *D, int64_t ld)
{
    for (int64_t i = 0; i < n; ++i) {
        D[i * ld + i] = 0.0f;
        for (int64_t j = 0; j < i; ++j) {
            float v = icl_ref_ward_distance(C + i * d, sizes ? sizes[i] : 1, C + j * d, sizes ? sizes[j] : 1, d);
            D[i * ld + j] = v;
            D[j * ld + i] = v;
        }
    }
}

/* clustering.go:119-133 FindClosestClusters: strict '<' against MaxFloat32, row-major lower triangle. */
void icl_ref_find_closest(const float *D, int64_t n, int64_t ld, int64_t *oi, int64_t *oj)
{
    float min = ICL_REF_MAXF;
    int64_t bi = -1, bj = -1;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < i; ++j)
            if (D[i * ld + j] < min) {
                min = D[i * ld + j];
                bi = i;
                bj = j;
            }
    *oi = bi;
    *oj = bj;
}

/* clustering.go:168-186 CalculateOptimalClusters. Returns 0 and *k on success, nonzero on the
 * reference's two error branches.  minSize<1 or maxSize<1 are rejected (division by zero in the
 * reference yields an implementation-defined int conversion). */
int icl_ref_calc_optimal_clusters(int64_t total, int64_t min_size, int64_t max_size, int64_t *k)
{
    if (min_size < 1 || max_size < 1) return 3;
    if (total < min_size) return 1; /* :169 */
    int64_t lo = (int64_t)ceil((double)total / (double)max_size);  /* :173 */
    int64_t hi = (int64_t)floor((double)total / (double)min_size); /* :174 */
    if (lo > hi) return 2;                                         /* :175 */
    int64_t n = lo;
    if (lo < hi) n = (lo + hi) / 2; /* :181-183 */
    *k = n;
    return 0;
}

typedef struct {
    int64_t *indices; /* clustering.go:12 */
    int64_t nidx;
    int64_t size;     /* :13 */
    float *centroid;  /* :14 */
    int64_t cid;      /* creation id (not in the reference; bookkeeping for the merge log only) */
} ref_cluster;

/*
 * clustering.go:198-284 PerformClusteringWithConstraints on flat E[n][d].
 * Outputs (canonical comparable form):
 *   cluster_id[n]  dense id of the kept cluster holding image i, -1 if its cluster was dropped (<minSize)
 *   member_rank[n] position of image i inside its cluster's member list, -1 if dropped
 *   *n_clusters    number of kept clusters
 *   merge_log      optional, 4 int64 per performed merge: pos_i, pos_j, creation_id_i, creation_id_j
 *   *n_merges, *n_skips optional counters (merges performed, oversize bans :228-234)
 * Returns 0 on success ("true"), 1/2/3 for the constraint errors ("nil,false"), 4 if an oversize
 * cluster is ever observed (unreachable, see SURVEY.md 8a C10: splitCluster is not restated).
 */
int icl_ref_cluster(const float *E, int64_t n, int64_t d, int64_t min_size, int64_t max_size,
                    int32_t *cluster_id, int32_t *member_rank, int32_t *n_clusters,
                    int64_t *merge_log, int64_t *n_merges, int64_t *n_skips)
{
    int64_t k = 0;
    int rc = icl_ref_calc_optimal_clusters(n, min_size, max_size, &k); /* :203 */
    if (rc) return rc;

    ref_cluster *cl = (ref_cluster *)malloc((size_t)(n > 0 ? n : 1) * sizeof(ref_cluster));
    for (int64_t i = 0; i < n; ++i) { /* :211-214 NewCluster */
        cl[i].indices = (int64_t *)malloc(sizeof(int64_t));
        cl[i].indices[0] = i;
        cl[i].nidx = 1;
        cl[i].size = 1;
        cl[i].centroid = (float *)malloc((size_t)d * sizeof(float));
        memcpy(cl[i].centroid, E + i * d, (size_t)d * sizeof(float));
        cl[i].cid = i;
    }
    int64_t len = n;

    /* :217 distance matrix as separately allocated rows, like [][]float32 */
    float **D = (float **)malloc((size_t)(n > 0 ? n : 1) * sizeof(float *));
    for (int64_t i = 0; i < n; ++i) {
        D[i] = (float *)calloc((size_t)n, sizeof(float));
    }
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < i; ++j) {
            float v = icl_ref_ward_distance(cl[i].centroid, cl[i].size, cl[j].centroid, cl[j].size, d);
            D[i][j] = v;
            D[j][i] = v;
        }

    int64_t merges = 0, skips = 0, next_cid = n;
    while (len > k) { /* :220 */
        /* :221 FindClosestClusters */
        float min = ICL_REF_MAXF;
        int64_t bi = -1, bj = -1;
        for (int64_t i = 0; i < len; ++i) {
            const float *row = D[i];
            for (int64_t j = 0; j < i; ++j)
                if (row[j] < min) {
                    min = row[j];
                    bi = i;
                    bj = j;
                }
        }
        if (bi == -1 || bj == -1) break; /* :222-225 */

        if (cl[bi].size + cl[bj].size > max_size) { /* :228-234 */
            D[bi][bj] = ICL_REF_MAXF;
            D[bj][bi] = ICL_REF_MAXF;
            ++skips;
            continue;
        }

        /* :237 MergeClusters(clusters[i], clusters[j]) : a = higher position */
        ref_cluster a = cl[bi], b = cl[bj], nc;
        nc.nidx = a.nidx + b.nidx;
        nc.indices = (int64_t *)malloc((size_t)nc.nidx * sizeof(int64_t));
        memcpy(nc.indices, a.indices, (size_t)a.nidx * sizeof(int64_t));            /* :31 a first */
        memcpy(nc.indices + a.nidx, b.indices, (size_t)b.nidx * sizeof(int64_t));
        nc.size = a.size + b.size;                                                  /* :34 */
        nc.centroid = (float *)malloc((size_t)d * sizeof(float));
        icl_ref_merge_centroid(a.centroid, a.size, b.centroid, b.size, d, nc.centroid); /* :37-40 */
        nc.cid = next_cid++;
        if (merge_log) {
            merge_log[4 * merges + 0] = bi;
            merge_log[4 * merges + 1] = bj;
            merge_log[4 * merges + 2] = a.cid;
            merge_log[4 * merges + 3] = b.cid;
        }
        ++merges;

        /* :240 RemoveClusters (order preserving; bj < bi) then :241 append */
        int64_t lo = bj, hi = bi;
        free(a.indices); free(a.centroid); free(b.indices); free(b.centroid);
        memmove(cl + hi, cl + hi + 1, (size_t)(len - hi - 1) * sizeof(ref_cluster));
        memmove(cl + lo, cl + lo + 1, (size_t)(len - 1 - lo - 1) * sizeof(ref_cluster));
        len -= 2;
        cl[len++] = nc;

        /* :244 UpdateDistanceMatrix -> :100-116 RemoveRowsAndColumns (old length = len+1) */
        int64_t oldn = len + 1;
        for (int64_t r = 0; r < oldn; ++r) {
            float *row = D[r];
            memmove(row + hi, row + hi + 1, (size_t)(oldn - hi - 1) * sizeof(float));
            memmove(row + lo, row + lo + 1, (size_t)(oldn - 1 - lo - 1) * sizeof(float));
        }
        float *rhi = D[hi], *rlo = D[lo];
        memmove(D + hi, D + hi + 1, (size_t)(oldn - hi - 1) * sizeof(float *));
        memmove(D + lo, D + lo + 1, (size_t)(oldn - 1 - lo - 1) * sizeof(float *));
        free(rhi);
        /* :81-93 new row from centroids (NOT Lance-Williams); reuse rlo's storage for it */
        float *new_row = rlo;
        for (int64_t i2 = 0; i2 < len - 1; ++i2) {
            float v = icl_ref_ward_distance(cl[i2].centroid, cl[i2].size, nc.centroid, nc.size, d); /* :84 */
            new_row[i2] = v;
        }
        new_row[len - 1] = 0.0f; /* :87 */
        for (int64_t i2 = 0; i2 < len - 1; ++i2) D[i2][len - 1] = new_row[i2]; /* :90-92 */
        D[len - 1] = new_row;                                                   /* :93 */
    }

    /* :249-262 oversize handling: unreachable for max_size >= 1; flag instead of restating splitCluster */
    int ret = 0;
    for (int64_t c = 0; c < len; ++c)
        if (cl[c].size > max_size) ret = 4;

    /* :265-280 dense ids in position order, dropping clusters below min_size */
    for (int64_t i = 0; i < n; ++i) {
        cluster_id[i] = -1;
        member_rank[i] = -1;
    }
    int32_t cid = 0;
    for (int64_t c = 0; c < len; ++c) {
        if (cl[c].size < min_size) continue; /* :268-271 */
        for (int64_t r = 0; r < cl[c].nidx; ++r) {
            cluster_id[cl[c].indices[r]] = cid;
            member_rank[cl[c].indices[r]] = (int32_t)r;
        }
        ++cid;
    }
    *n_clusters = cid;
    if (n_merges) *n_merges = merges;
    if (n_skips) *n_skips = skips;

    for (int64_t c = 0; c < len; ++c) {
        free(cl[c].indices);
        free(cl[c].centroid);
        free(D[c]);
    }
    free(cl);
    free(D);
    return ret;
}
