/*
 * tgp_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, never shipped, never on the product path)
 * of the integer/bit-sensitive pieces of the TG-Pose point-cloud hot path:
 *
 *   - brute-force kNN index build        (reference network/fs_net_repo/gcn3d.py:14-23)
 *   - 1-nearest index for up-sampling    (reference network/fs_net_repo/gcn3d.py:26-35)
 *   - Chamfer nearest-neighbour search   (reference tools/pyTorchChamferDistance/chamfer_distance.cpp:59-87,
 *                                         CUDA twin losses/chamfer3D/chamfer3D.cu:12-134)
 *   - Chamfer backward                   (reference tools/pyTorchChamferDistance/chamfer_distance.cpp:114-177)
 *
 * The reference computes the kNN distances with torch ops whose fp32 rounding decides the
 * neighbour ORDER, so this file pins the arithmetic explicitly (measured against the imported
 * reference in the build container, torch 2.10 CPU / MKL, see DESIGN.md "Oracle"):
 *
 *   inner_ij = bmm(v, v^T)            -> ascending-k fused-multiply-add chain starting from 0
 *                                        (bit-identical to MKL sgemm here; also what
 *                                        v_mfma_f32_32x32x2_f32 computes on gfx950)
 *   q_i      = torch.sum(v**2, dim=2) -> squares rounded to fp32, then ATen's cascade sum
 *                                        (aten/src/ATen/native/cpu/SumKernel.cpp): 8-wide
 *                                        vectors, 4 interleaved accumulators, sequential lanes
 *   D_ij     = inner*(-2) + q[None,:] + q[:,None]   -> fl(fl(fl(-2*inner) + q_j) + q_i)
 *   topk(k+1, smallest, sorted)[:, 1:]              -> order by (D, j) ascending, drop rank 0.
 *                                                      torch leaves ties undefined; the lowest
 *                                                      index first rule is this build's policy.
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -ffp-contract=off -fopenmp).  -ffp-contract=off is
 * REQUIRED: every rounding below is intentional.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TGPO_VEC 8   /* at::vec::Vectorized<float>::size() as compiled into SumKernel (AVX2 width) */
#define TGPO_ILP 4   /* ilp_factor in row_sum() */
#define TGPO_LEVELS 4

/* ATen multi_row_sum<acc_t, 4>: `size` rows of 4 "columns", each column W floats wide. */
static void multi_row_sum(const float *in, long row_stride, long col_stride, long size, int W,
                          float out[TGPO_ILP][TGPO_VEC])
{
    float acc[TGPO_LEVELS][TGPO_ILP][TGPO_VEC];
    memset(acc, 0, sizeof(acc));
    const long level_power = 4; /* max(4, CeilLog2(size)/4) == 4 for every size < 2^20 */
    const long level_step = 1L << level_power;
    const long level_mask = level_step - 1;
    long i = 0;
    for (; i + level_step <= size;) {
        for (long j = 0; j < level_step; ++j, ++i)
            for (int k = 0; k < TGPO_ILP; ++k)
                for (int l = 0; l < W; ++l)
                    acc[0][k][l] = acc[0][k][l] + in[i * row_stride + k * col_stride + l];
        for (int j = 1; j < TGPO_LEVELS; ++j) {
            for (int k = 0; k < TGPO_ILP; ++k)
                for (int l = 0; l < W; ++l) {
                    acc[j][k][l] = acc[j][k][l] + acc[j - 1][k][l];
                    acc[j - 1][k][l] = 0.f;
                }
            const long mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; ++i)
        for (int k = 0; k < TGPO_ILP; ++k)
            for (int l = 0; l < W; ++l)
                acc[0][k][l] = acc[0][k][l] + in[i * row_stride + k * col_stride + l];
    for (int j = 1; j < TGPO_LEVELS; ++j)
        for (int k = 0; k < TGPO_ILP; ++k)
            for (int l = 0; l < W; ++l)
                acc[0][k][l] = acc[0][k][l] + acc[j][k][l];
    for (int k = 0; k < TGPO_ILP; ++k)
        for (int l = 0; l < W; ++l) out[k][l] = acc[0][k][l];
}

/* ATen row_sum<acc_t>: `size` elements (each W floats wide, stride W) */
static void row_sum(const float *in, long size, int W, float out[TGPO_VEC])
{
    float part[TGPO_ILP][TGPO_VEC];
    const long size_ilp = size / TGPO_ILP;
    multi_row_sum(in, (long)W * TGPO_ILP, W, size_ilp, W, part);
    for (long i = size_ilp * TGPO_ILP; i < size; ++i)
        for (int l = 0; l < W; ++l) part[0][l] = part[0][l] + in[i * W + l];
    for (int k = 1; k < TGPO_ILP; ++k)
        for (int l = 0; l < W; ++l) part[0][l] = part[0][l] + part[k][l];
    for (int l = 0; l < W; ++l) out[l] = part[0][l];
}

/* torch.sum over a contiguous inner dimension of length d (cascade_sum, SumKernel.cpp) */
static float torch_inner_sum(const float *x, int d)
{
    float lanes[TGPO_VEC];
    if (d >= TGPO_VEC) { /* vectorized_inner_sum */
        const long vec_size = d / TGPO_VEC;
        row_sum(x, vec_size, TGPO_VEC, lanes);
        float fin = 0.f;
        for (long k = vec_size * TGPO_VEC; k < d; ++k) fin = fin + x[k];
        for (int l = 0; l < TGPO_VEC; ++l) fin = fin + lanes[l];
        return 0.f + fin;
    }
    row_sum(x, d, 1, lanes); /* scalar_inner_sum */
    return 0.f + lanes[0];
}

/* q[r] = torch.sum(x[r]**2) for `rows` rows of length d.  gcn3d.py:19 / :32-33 */
void tgpo_sqnorm(const float *x, long rows, int d, float *q)
{
#pragma omp parallel
    {
        float *sq = (float *)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(static)
        for (long r = 0; r < rows; ++r) {
            for (int k = 0; k < d; ++k) sq[k] = x[r * d + k] * x[r * d + k];
            q[r] = torch_inner_sum(sq, d);
        }
        free(sq);
    }
}

static inline float inner_chain(const float *a, const float *b, int d)
{
    float acc = 0.f;
    for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

/* keep the kk smallest (dist, index) pairs in ascending order; ties -> lower index first */
static inline void topk_insert(float *bd, int *bi, int kk, int *cnt, float dv, int j)
{
    int pos;
    if (*cnt < kk) {
        pos = (*cnt)++;
    } else {
        if (!(dv < bd[kk - 1])) return; /* equal distance, larger index: stays out */
        pos = kk - 1;
    }
    while (pos > 0 && dv < bd[pos - 1]) { /* strict: an equal earlier (smaller j) entry stays ahead */
        bd[pos] = bd[pos - 1];
        bi[pos] = bi[pos - 1];
        --pos;
    }
    bd[pos] = dv;
    bi[pos] = j;
}

/*
 * get_neighbor_index (gcn3d.py:14-23).  x: (B,n,d) fp32.  idx: (B,n,k) int32.
 * dist (optional, may be NULL): (B,n,k) the distances of the kept neighbours.
 * first (optional): (B,n) the index that was dropped as rank 0 (normally the point itself).
 */
void tgpo_knn(const float *x, int B, int n, int d, int k, int32_t *idx, float *dist, int32_t *first)
{
    float *q = (float *)malloc(sizeof(float) * (size_t)B * n);
    tgpo_sqnorm(x, (long)B * n, d, q);
    const int kk = k + 1;
#pragma omp parallel
    {
        float *bd = (float *)malloc(sizeof(float) * kk);
        int *bi = (int *)malloc(sizeof(int) * kk);
#pragma omp for schedule(dynamic, 16)
        for (long r = 0; r < (long)B * n; ++r) {
            const int b = (int)(r / n);
            const float *xb = x + (size_t)b * n * d;
            const float *qb = q + (size_t)b * n;
            const float *xi = x + (size_t)r * d;
            const float qi = q[r];
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                const float inner = inner_chain(xi, xb + (size_t)j * d, d);
                const float t1 = inner * -2.0f;
                const float t2 = t1 + qb[j];
                const float dv = t2 + qi;
                topk_insert(bd, bi, kk, &cnt, dv, j);
            }
            for (int t = 0; t < k; ++t) {
                idx[r * k + t] = bi[t + 1];
                if (dist) dist[r * k + t] = bd[t + 1];
            }
            if (first) first[r] = bi[0];
        }
        free(bd);
        free(bi);
    }
    free(q);
}

/*
 * get_nearest_index (gcn3d.py:26-35): d = s_norm[None,:] + t_norm[:,None] - 2*inner, topk k=1.
 * tgt: (B,n,d), src: (B,m,d) -> idx (B,n) int32 (lowest index on ties).
 */
void tgpo_nn1(const float *tgt, const float *src, int B, int n, int m, int d, int32_t *idx)
{
    float *qt = (float *)malloc(sizeof(float) * (size_t)B * n);
    float *qs = (float *)malloc(sizeof(float) * (size_t)B * m);
    tgpo_sqnorm(tgt, (long)B * n, d, qt);
    tgpo_sqnorm(src, (long)B * m, d, qs);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)B * n; ++r) {
        const int b = (int)(r / n);
        const float *ti = tgt + (size_t)r * d;
        float best = 0.f;
        int besti = 0;
        for (int j = 0; j < m; ++j) {
            const float inner = inner_chain(ti, src + ((size_t)b * m + j) * d, d);
            const float s = qs[(size_t)b * m + j] + qt[r];
            const float dv = s - 2.0f * inner;
            if (j == 0 || dv < best) {
                best = dv;
                besti = j;
            }
        }
        idx[r] = besti;
    }
    free(qt);
    free(qs);
}

/* one direction of the Chamfer search: chamfer_distance.cpp:59-87 (float products, first-min wins) */
static void nnsearch(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int32_t *idx)
{
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)b * n; ++r) {
        const int i = (int)(r / n);
        const float x1 = xyz1[r * 3 + 0], y1 = xyz1[r * 3 + 1], z1 = xyz1[r * 3 + 2];
        float best = 0.f;
        int besti = 0;
        for (int k = 0; k < m; ++k) {
            const float x2 = xyz2[((size_t)i * m + k) * 3 + 0] - x1;
            const float y2 = xyz2[((size_t)i * m + k) * 3 + 1] - y1;
            const float z2 = xyz2[((size_t)i * m + k) * 3 + 2] - z1;
            const float d = (x2 * x2 + y2 * y2) + z2 * z2; /* chamfer3D.cu:30-33, no contraction */
            if (k == 0 || d < best) {
                best = d;
                besti = k;
            }
        }
        dist[r] = best;
        idx[r] = besti;
    }
}

/* chamfer_3D.forward (chamfer_cuda.cpp:17-19 -> chamfer3D.cu:136-152) */
void tgpo_chamfer_fwd(const float *xyz1, const float *xyz2, int B, int n, int m, float *dist1, float *dist2,
                      int32_t *idx1, int32_t *idx2)
{
    nnsearch(B, n, m, xyz1, xyz2, dist1, idx1);
    nnsearch(B, m, n, xyz2, xyz1, dist2, idx2);
}

/*
 * chamfer_3D.backward: accumulates into grad_xyz1/grad_xyz2 (caller zero-fills, dist_chamfer_3D.py:56-60).
 * Summation order is the serial CPU order of chamfer_distance.cpp:140-175: per batch element first
 * the xyz1 loop (own term into grad1, scattered term into grad2), then the xyz2 loop.
 */
void tgpo_chamfer_bwd(const float *xyz1, const float *xyz2, int B, int n, int m, const float *gd1,
                      const float *gd2, const int32_t *idx1, const int32_t *idx2, float *g1, float *g2)
{
    for (int i = 0; i < B; ++i) {
        for (int j = 0; j < n; ++j) {
            const size_t a = ((size_t)i * n + j) * 3;
            const int j2 = idx1[(size_t)i * n + j];
            const size_t c = ((size_t)i * m + j2) * 3;
            const float g = gd1[(size_t)i * n + j] * 2.0f;
            for (int t = 0; t < 3; ++t) {
                const float v = g * (xyz1[a + t] - xyz2[c + t]);
                g1[a + t] = g1[a + t] + v;
                g2[c + t] = g2[c + t] - v;
            }
        }
        for (int j = 0; j < m; ++j) {
            const size_t a = ((size_t)i * m + j) * 3;
            const int j2 = idx2[(size_t)i * m + j];
            const size_t c = ((size_t)i * n + j2) * 3;
            const float g = gd2[(size_t)i * m + j] * 2.0f;
            for (int t = 0; t < 3; ++t) {
                const float v = g * (xyz2[a + t] - xyz1[c + t]);
                g2[a + t] = g2[a + t] + v;
                g1[c + t] = g1[c + t] - v;
            }
        }
    }
}

/* plain pairwise distance matrix for one batch element, exposed for tests of the definition itself */
void tgpo_knn_dist_matrix(const float *x, int n, int d, float *D)
{
    float *q = (float *)malloc(sizeof(float) * (size_t)n);
    tgpo_sqnorm(x, n, d, q);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const float inner = inner_chain(x + (size_t)i * d, x + (size_t)j * d, d);
            const float t1 = inner * -2.0f;
            const float t2 = t1 + q[j];
            D[(size_t)i * n + j] = t2 + q[i];
        }
    free(q);
}
