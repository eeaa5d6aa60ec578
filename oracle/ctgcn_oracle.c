/*
 * ctgcn_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's arithmetic for the CTGCN hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; nothing under ctgcn_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py)
 * against vectors produced by running the reference itself in the build container
 * (tests/golden/make_golden.py): core numbers from networkx.core_number via the
 * reference's get_nx_graph, the reference loader's matrices, and the reference
 * CoreDiffusion aggregation.  The reference's own tests hold no vectors for this
 * path.
 *
 * Third-party arithmetic restated here:
 *   - networkx.algorithms.core.core_number (pinned 2.8.8 in the reference's poetry.lock,
 *     2.4 in its Dockerfile): Batagelj–Zaversnik O(m) bin-sort peel on unweighted degree.
 *     Call site: preprocessing/structure_generation.py:35.
 *   - torch.sparse.mm (ATen CPU COO/CSR SpMM), call sites layers.py:43,45: coalesce
 *     (row-major, column-sorted), then per row a sequential sum over the row's entries.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------
 * core numbers.  Follows networkx core_number as called at structure_generation.py:35 on the
 * graph built by utils.py:23-30 (undirected, simple, self loops removed).  Input is that
 * graph's symmetric CSR structure (indptr/indices), values ignored (unweighted degree).
 * The result is unique, so any correct peel must agree bit for bit.
 * ------------------------------------------------------------------------------------- */
int oracle_kcore_bz(int64_t n, const int64_t *indptr, const int32_t *indices, int32_t *core)
{
    if (n == 0) return 0;
    int32_t *deg = core; /* peeled in place */
    int32_t maxdeg = 0;
    for (int64_t v = 0; v < n; ++v) {
        int64_t dv = 0;
        for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e)
            if (indices[e] != v) ++dv; /* a self loop never counts */
        deg[v] = (int32_t)dv;
        if (deg[v] > maxdeg) maxdeg = deg[v];
    }
    int64_t *bin = (int64_t *)calloc((size_t)maxdeg + 2, sizeof(int64_t));
    int64_t *pos = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int32_t *vert = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    if (!bin || !pos || !vert) { free(bin); free(pos); free(vert); return -1; }
    for (int64_t v = 0; v < n; ++v) bin[deg[v]]++;
    int64_t start = 0;
    for (int32_t d = 0; d <= maxdeg; ++d) { int64_t c = bin[d]; bin[d] = start; start += c; }
    for (int64_t v = 0; v < n; ++v) { pos[v] = bin[deg[v]]; vert[pos[v]] = (int32_t)v; bin[deg[v]]++; }
    for (int32_t d = maxdeg; d >= 1; --d) bin[d] = bin[d - 1];
    bin[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t v = vert[i];
        for (int64_t e = indptr[v]; e < indptr[v + 1]; ++e) {
            int32_t u = indices[e];
            if (u == v) continue;
            if (deg[u] > deg[v]) {
                int32_t du = deg[u];
                int64_t pu = pos[u], pw = bin[du];
                int32_t w = vert[pw];
                if (u != w) { pos[u] = pw; vert[pu] = w; pos[w] = pu; vert[pw] = u; }
                bin[du]++;
                deg[u]--;
            }
        }
    }
    free(bin); free(pos); free(vert);
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * Y = A·X (or Y += A·X), A in canonical CSR (what torch's coalesce yields), fp32.
 * Restates torch.sparse.mm(adj, x) at layers.py:43,45.
 * ------------------------------------------------------------------------------------- */
int oracle_spmm_csr_f32(int64_t n, int64_t d, const int64_t *indptr, const int32_t *indices,
                        const float *val, const float *X, float *Y, int accumulate)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n; ++r) {
        float *y = Y + r * d;
        if (!accumulate) memset(y, 0, (size_t)d * sizeof(float));
        for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) {
            const float a = val[e];
            const float *x = X + (int64_t)indices[e] * d;
            for (int64_t c = 0; c < d; ++c) y[c] += a * x[c];
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * CoreDiffusion aggregation, layers.py:41-48 + the stack/transpose of :58.
 *   res_0 = A_0 X ; res_j = res_{j-1} + A_j X ; H[:, j, :] = relu(res_j)
 * K independent CSR matrices (as the reference keeps them); H is [n, K, d] row-major.
 * `pre` (optional, may be NULL) receives the un-rectified res_j in the same layout.
 * ------------------------------------------------------------------------------------- */
int oracle_core_aggregate_f32(int64_t n, int64_t d, int32_t K, const int64_t *const *indptr,
                              const int32_t *const *indices, const float *const *val,
                              const float *X, float *H, float *pre)
{
    float *res = (float *)malloc((size_t)n * d * sizeof(float));
    float *tmp = (float *)malloc((size_t)n * d * sizeof(float));
    if (!res || !tmp) { free(res); free(tmp); return -1; }
    for (int32_t j = 0; j < K; ++j) {
        oracle_spmm_csr_f32(n, d, indptr[j], indices[j], val[j], X, tmp, 0);
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n; ++r) {
            float *rr = res + r * d;
            const float *t = tmp + r * d;
            float *h = H + (r * K + j) * d;
            for (int64_t c = 0; c < d; ++c) {
                float v = (j == 0) ? t[c] : rr[c] + t[c];
                rr[c] = v;
                if (pre) pre[(r * K + j) * d + c] = v;
                h[c] = v > 0.0f ? v : 0.0f;
            }
        }
    }
    free(res); free(tmp);
    return 0;
}

/* gradient of sum(H * dH) w.r.t. X for the aggregation above (autograd of layers.py:41-48):
 *   G_j = dH_j * [res_j > 0] ; dX = sum_j (sum_{i<=j} A_i)^T G_j = sum_i A_i^T (sum_{j>=i} G_j)
 * Transposes are taken explicitly (no symmetry assumption). */
int oracle_core_aggregate_bwd_f32(int64_t n, int64_t d, int32_t K, const int64_t *const *indptr,
                                  const int32_t *const *indices, const float *const *val,
                                  const float *pre, const float *dH, float *dX)
{
    float *S = (float *)calloc((size_t)n * d, sizeof(float));
    if (!S) return -1;
    memset(dX, 0, (size_t)n * d * sizeof(float));
    for (int32_t i = K - 1; i >= 0; --i) {
        for (int64_t r = 0; r < n; ++r)
            for (int64_t c = 0; c < d; ++c) {
                int64_t o = (r * K + i) * d + c;
                if (pre[o] > 0.0f) S[r * d + c] += dH[o];
            }
        /* dX += A_i^T S : scatter along the rows of A_i */
        for (int64_t r = 0; r < n; ++r)
            for (int64_t e = indptr[i][r]; e < indptr[i][r + 1]; ++e) {
                const float a = val[i][e];
                float *o = dX + (int64_t)indices[i][e] * d;
                const float *s = S + r * d;
                for (int64_t c = 0; c < d; ++c) o[c] += a * s[c];
            }
    }
    free(S);
    return 0;
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}
