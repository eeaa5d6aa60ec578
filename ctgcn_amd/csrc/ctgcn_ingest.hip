// ctgcn_ingest.hip — edge rows -> symmetric, de-duplicated, zero-diagonal CSR on the GPU (gfx950).
// Replaces the graph construction of the reference's utils.py:23-30 (get_nx_graph: nx.from_pandas_edgelist
// + remove self loops) and utils.py:35-58 (get_sp_adj_mat: A[i,j] = A[j,i] = w overwrite loop):
//   every row sets weight({src,dst}) = w; the LAST row naming an unordered pair wins; src == dst rows are dropped.
// Integer/byte work, HBM bound: two rocPRIM radix sorts (stable) + three small kernels.  Part of libctgcn_hip.so.
#include <cstdint>
#include <cstdio>
#include <cstring>   // rocprim's texture iterator calls the host memset

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "../../include/ctgcn_hip.h"

extern "C" int ctgcn_set_error_(int code, const char *msg);   // defined in ctgcn_hip.hip

namespace {

#define ING_TRY(expr)                                                                \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            char buf[384];                                                           \
            snprintf(buf, sizeof(buf), "%s -> %s", #expr, hipGetErrorString(e_));   \
            return ctgcn_set_error_(CTGCN_E_HIP, buf);                               \
        }                                                                            \
    } while (0)

constexpr uint64_t DROPPED = ~0ull;

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__global__ void pair_keys_kernel(int64_t m, uint64_t n, const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                 uint64_t *__restrict__ key, uint32_t *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const uint64_t s = (uint32_t)src[i], d = (uint32_t)dst[i];
    key[i] = (s == d || s >= n || d >= n) ? DROPPED : (s < d ? s * n + d : d * n + s);
    idx[i] = (uint32_t)i;
}

// after the stable sort the last element of every run of equal keys is the winning row
__global__ void mark_winners_kernel(int64_t m, const uint64_t *__restrict__ key, uint32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const uint64_t k = key[i];
    flag[i] = (k != DROPPED && (i + 1 == m || key[i + 1] != k)) ? 1u : 0u;
}

// winner p (pair lo<hi, weight w) -> two directed entries keyed row*n+col
__global__ void emit_entries_kernel(int64_t m, uint64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ idx,
                                    const uint32_t *__restrict__ flag, const uint32_t *__restrict__ pos,
                                    const float *__restrict__ w, uint64_t *__restrict__ ekey, float *__restrict__ eval)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m || !flag[i]) return;
    const uint64_t k = key[i], lo = k / n, hi = k % n;
    const float ww = w ? w[idx[i]] : 1.0f;
    const uint64_t p = pos[i];
    ekey[2 * p] = lo * n + hi;
    ekey[2 * p + 1] = hi * n + lo;
    eval[2 * p] = ww;
    eval[2 * p + 1] = ww;
}

__global__ void finish_csr_kernel(int64_t n, int64_t nnz, uint64_t nn, const uint64_t *__restrict__ ekey,
                                  int32_t *__restrict__ row_ptr, int32_t *__restrict__ col)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nnz) col[i] = (int32_t)(ekey[i] % nn);
    if (i <= n) {   // row_ptr[i] = first entry whose key >= i*n
        const uint64_t target = (uint64_t)i * nn;
        int64_t lo = 0, hi = nnz;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (ekey[mid] < target) lo = mid + 1; else hi = mid;
        }
        row_ptr[i] = (int32_t)lo;
    }
}

int key_bits(uint64_t n)
{
    int b = 1;
    while (b < 32 && (1ull << b) < n) ++b;
    return 2 * b > 64 ? 64 : 2 * b;
}

struct Layout {
    size_t key_a, key_b, idx_a, idx_b, flag, pos, ekey_a, ekey_b, eval_a, eval_b, tmp, tmp_bytes, total;
};

int plan(int64_t n, int64_t m, Layout &L)
{
    size_t s1 = 0, s2 = 0, s3 = 0;
    const int bits = key_bits((uint64_t)n);
    if (rocprim::radix_sort_pairs(nullptr, s1, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                  (size_t)m, 0, 64, 0) != hipSuccess) return -1;
    if (rocprim::radix_sort_pairs(nullptr, s2, (uint64_t *)nullptr, (uint64_t *)nullptr, (float *)nullptr, (float *)nullptr,
                                  (size_t)(2 * m), 0, bits, 0) != hipSuccess) return -1;
    if (rocprim::exclusive_scan(nullptr, s3, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)m, rocprim::plus<uint32_t>(), 0) != hipSuccess) return -1;
    L.tmp_bytes = s1 > s2 ? s1 : s2;
    if (s3 > L.tmp_bytes) L.tmp_bytes = s3;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    const size_t mm = (size_t)(m > 0 ? m : 1);
    L.key_a = take(mm * 8); L.key_b = take(mm * 8); L.idx_a = take(mm * 4); L.idx_b = take(mm * 4);
    L.flag = take(mm * 4); L.pos = take(mm * 4);
    L.ekey_a = take(2 * mm * 8); L.ekey_b = take(2 * mm * 8); L.eval_a = take(2 * mm * 4); L.eval_b = take(2 * mm * 4);
    L.tmp = take(L.tmp_bytes);
    L.total = off;
    return 0;
}

}  // namespace

extern "C" size_t ctgcn_ingest_workspace_bytes_(int64_t n, int64_t m)
{
    Layout L{};
    if (m < 0 || n < 0 || plan(n, m, L) != 0) return 0;
    return L.total;
}

extern "C" int ctgcn_edges_to_csr(int64_t n, int64_t m, const int32_t *src, const int32_t *dst, const float *w,
                                  int32_t *row_ptr, int32_t *col_idx, float *val, int64_t *nnz_host,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || m < 0 || n > 0x7fffffffLL || 2 * m > 0x7fffffffLL) return ctgcn_set_error_(CTGCN_E_INVALID, "edges_to_csr: n or 2*m out of int32 range");
    if (!row_ptr || !nnz_host || (m > 0 && (!src || !dst || !col_idx || !val || !workspace)))
        return ctgcn_set_error_(CTGCN_E_INVALID, "edges_to_csr: null pointer");
    hipStream_t st = (hipStream_t)stream;
    *nnz_host = 0;
    if (m == 0) {
        ING_TRY(hipMemsetAsync(row_ptr, 0, (size_t)(n + 1) * 4, st));
        return CTGCN_OK;
    }
    Layout L{};
    if (plan(n, m, L) != 0) return ctgcn_set_error_(CTGCN_E_HIP, "edges_to_csr: rocprim size query failed");
    if (workspace_bytes < L.total) return ctgcn_set_error_(CTGCN_E_WORKSPACE, "edges_to_csr: workspace too small");
    char *ws = (char *)workspace;
    uint64_t *key_a = (uint64_t *)(ws + L.key_a), *key_b = (uint64_t *)(ws + L.key_b);
    uint32_t *idx_a = (uint32_t *)(ws + L.idx_a), *idx_b = (uint32_t *)(ws + L.idx_b);
    uint32_t *flag = (uint32_t *)(ws + L.flag), *pos = (uint32_t *)(ws + L.pos);
    uint64_t *ekey_a = (uint64_t *)(ws + L.ekey_a), *ekey_b = (uint64_t *)(ws + L.ekey_b);
    float *eval_a = (float *)(ws + L.eval_a);
    void *tmp = ws + L.tmp;
    size_t tmp_bytes = L.tmp_bytes;
    const unsigned blocks_m = (unsigned)((m + 255) / 256);

    hipLaunchKernelGGL(pair_keys_kernel, dim3(blocks_m), dim3(256), 0, st, m, (uint64_t)n, src, dst, key_a, idx_a);
    // stable: equal keys keep file order, so the last of a run is the last row of the file
    ING_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, key_a, key_b, idx_a, idx_b, (size_t)m, 0, 64, st));
    hipLaunchKernelGGL(mark_winners_kernel, dim3(blocks_m), dim3(256), 0, st, m, key_b, flag);
    tmp_bytes = L.tmp_bytes;
    ING_TRY(rocprim::exclusive_scan(tmp, tmp_bytes, flag, pos, 0u, (size_t)m, rocprim::plus<uint32_t>(), st));
    uint32_t last_pos = 0, last_flag = 0;
    ING_TRY(hipMemcpyAsync(&last_pos, pos + (m - 1), 4, hipMemcpyDeviceToHost, st));
    ING_TRY(hipMemcpyAsync(&last_flag, flag + (m - 1), 4, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(emit_entries_kernel, dim3(blocks_m), dim3(256), 0, st, m, (uint64_t)n, key_b, idx_b, flag, pos, w, ekey_a, eval_a);
    ING_TRY(hipStreamSynchronize(st));
    const int64_t nnz = 2 * (int64_t)(last_pos + last_flag);
    *nnz_host = nnz;
    if (nnz > 0) {
        tmp_bytes = L.tmp_bytes;
        ING_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, ekey_a, ekey_b, eval_a, val, (size_t)nnz, 0, key_bits((uint64_t)n), st));
    }
    const int64_t work = (nnz > n + 1 ? nnz : n + 1);
    hipLaunchKernelGGL(finish_csr_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, n, nnz, (uint64_t)n, ekey_b, row_ptr, col_idx);
    ING_TRY(hipGetLastError());
    return CTGCN_OK;
}
