// ctgcn_walks.hip — random-walk corpus and negative-sampling index draws on the GPU.
// Replaces the Python loops of the reference's preprocessing/random_walk.py:8-69 (weighted random walks, co-occurrence
// pairs, node frequencies) and metrics.py:62-93 (per-batch-node positive draws, shared negative draws).  Integer /
// index work; the draws themselves are random by specification (the reference reseeds from OS entropy, metrics.py:66),
// so parity is structural: every emitted pair is a pair the reference could emit, frequencies count the same events,
// draws are uniform without replacement.  Counter-based RNG (splitmix64 of seed/node/walk/step): reproducible per seed.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../include/ctgcn_hip.h"

extern "C" int ctgcn_set_error_(int code, const char *msg);

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t a, uint64_t b, uint64_t c)
{
    return (double)(mix64(mix64(a) ^ mix64(b * 0x100000001b3ull + c)) >> 11) * (1.0 / 9007199254740992.0);
}

constexpr int MAX_WALK = 32;      // walk_length + 1 <= 32

// one thread per (node, walk).  cumw: per-row inclusive prefix sums of the edge weights (random_walk.py:32-35 normalises
// the weights of the current node; drawing u*rowsum in the prefix sums is the same distribution).
__global__ __launch_bounds__(256) void walk_kernel(int64_t n, int walk_len, int walk_time, int walk0, uint64_t seed, int weighted,
                                                   const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                   const float *__restrict__ cumw, int32_t *__restrict__ src_out,
                                                   int32_t *__restrict__ dst_out, unsigned long long *__restrict__ freq)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * walk_time) return;
    const int64_t node = gid / walk_time;
    const int it = (int)(gid % walk_time) + walk0;
    int walk[MAX_WALK];
    int len = 1;
    walk[0] = (int)node;
    while (len < walk_len) {
        const int cur = walk[len - 1];
        const int s = row_ptr[cur], e = row_ptr[cur + 1];
        if (e == s) break;                                              // random_walk.py:29-30: dead end
        const double u = u01(seed, (uint64_t)node * 1000003ull + (uint64_t)it, (uint64_t)len);
        int pick;
        if (weighted) {
            const float target = (float)(u * (double)cumw[e - 1]);
            int lo = s, hi = e - 1;                                     // first index with cumw > target
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cumw[mid] > target) hi = mid; else lo = mid + 1;
            }
            pick = lo;
        } else {
            pick = s + min((int)(u * (double)(e - s)), e - s - 1);
        }
        walk[len++] = col[pick];
    }
    // co-occurrence pairs (random_walk.py:40-50): every i < j with different endpoints
    const int per_walk = walk_len * (walk_len - 1) / 2;
    int64_t o = gid * per_walk;
    for (int i = 0; i < walk_len; ++i)
        for (int j = i + 1; j < walk_len; ++j, ++o) {
            if (i < len && j < len && walk[i] != walk[j]) {
                src_out[o] = walk[i];
                dst_out[o] = walk[j];
                atomicAdd(&freq[walk[i]], 1ull);
                atomicAdd(&freq[walk[j]], 1ull);
            } else {
                src_out[o] = 0;                                         // self pair: dropped by the de-duplicating ingest
                dst_out[o] = 0;
            }
        }
}

// per-row inclusive prefix sums of the weights (rows are short; one thread per row)
__global__ __launch_bounds__(256) void row_cumsum_kernel(int64_t n, const int32_t *__restrict__ row_ptr, const float *__restrict__ val,
                                                         float *__restrict__ cumw)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    float acc = 0.f;
    for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) { acc += val[e]; cumw[e] = acc; }
}

// metrics.py:70-84: for every batch node take all of its pair partners if there are at most `num`, else `num` of them
// drawn uniformly without replacement (selection sampling, order preserving); offsets = exclusive scan of min(deg,num).
__global__ __launch_bounds__(256) void pos_sample_kernel(int64_t batch, const int64_t *__restrict__ nodes, const int32_t *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ col, int num, uint64_t seed,
                                                         const int64_t *__restrict__ offsets, int64_t *__restrict__ node_out,
                                                         int64_t *__restrict__ pos_out)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const int64_t v = nodes[b];
    const int s = row_ptr[v], deg = row_ptr[v + 1] - s;
    int64_t o = offsets[b];
    int need = min(deg, num);
    for (int k = 0; k < deg && need > 0; ++k) {
        const bool take = (deg <= num) || (u01(seed, (uint64_t)b, (uint64_t)k) * (double)(deg - k) < (double)need);
        if (take) { node_out[o] = v; pos_out[o] = col[s + k]; ++o; --need; }
    }
}

// metrics.py:88: `num` distinct POSITIONS of the negative table (random.sample), one thread (num is ~20)
__global__ void neg_sample_kernel(int64_t table_len, const int32_t *__restrict__ table, int num, uint64_t seed, int64_t *__restrict__ neg_out,
                                  int64_t *__restrict__ pos_scratch)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int got = 0;
    for (uint64_t tries = 0; got < num; ++tries) {
        const int64_t p = min((int64_t)(u01(seed, 0x5eedull, tries) * (double)table_len), table_len - 1);
        bool dup = false;
        for (int i = 0; i < got; ++i) dup |= (pos_scratch[i] == p);
        if (!dup) { pos_scratch[got] = p; neg_out[got] = table[p]; ++got; }
    }
}

}  // namespace

#define WK_TRY(expr)                                                                 \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            char buf[384];                                                           \
            snprintf(buf, sizeof(buf), "%s -> %s", #expr, hipGetErrorString(e_));   \
            return ctgcn_set_error_(CTGCN_E_HIP, buf);                               \
        }                                                                            \
    } while (0)

extern "C" int ctgcn_row_cumsum_f32(int64_t n, const int32_t *row_ptr, const float *val, float *cumw, void *stream)
{
    if (n < 0) return ctgcn_set_error_(CTGCN_E_INVALID, "row_cumsum: bad size");
    if (n == 0) return CTGCN_OK;
    if (!row_ptr) return ctgcn_set_error_(CTGCN_E_INVALID, "row_cumsum: null pointer");
    hipLaunchKernelGGL(row_cumsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, row_ptr, val, cumw);
    WK_TRY(hipGetLastError());
    return CTGCN_OK;
}

extern "C" int ctgcn_random_walk_pairs(int64_t n, const int32_t *row_ptr, const int32_t *col_idx, const float *cumw,
                                       int32_t walk_length, int32_t walk_time, int32_t first_walk, uint64_t seed, int weighted,
                                       int32_t *pair_src, int32_t *pair_dst, int64_t *freq, void *stream)
{
    const int walk_len = walk_length + 1;                       // random_walk.py:11
    if (n < 0 || walk_length < 1 || walk_len > MAX_WALK || walk_time < 1) return ctgcn_set_error_(CTGCN_E_INVALID, "random_walk_pairs: bad sizes (walk_length in [1,31])");
    if (n == 0) return CTGCN_OK;
    if (!row_ptr || !pair_src || !pair_dst || !freq || (weighted && !cumw)) return ctgcn_set_error_(CTGCN_E_INVALID, "random_walk_pairs: null pointer");
    const int64_t threads = n * walk_time;
    if ((threads + 255) / 256 > 0x7fffffffLL) return ctgcn_set_error_(CTGCN_E_UNSUPPORTED, "random_walk_pairs: too many walks per call");
    hipLaunchKernelGGL(walk_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, walk_len, (int)walk_time,
                       (int)first_walk, seed, weighted, row_ptr, col_idx, cumw, pair_src, pair_dst, (unsigned long long *)freq);
    WK_TRY(hipGetLastError());
    return CTGCN_OK;
}

extern "C" int ctgcn_neg_sampling_indices(int64_t batch, const int64_t *batch_nodes, const int32_t *pair_row_ptr,
                                          const int32_t *pair_col, int32_t num, int64_t table_len, const int32_t *neg_table,
                                          uint64_t seed, const int64_t *offsets, int64_t *node_out, int64_t *pos_out,
                                          int64_t *neg_out, int64_t *scratch, void *stream)
{
    if (batch < 0 || num < 1 || table_len < num) return ctgcn_set_error_(CTGCN_E_INVALID, "neg_sampling_indices: bad sizes (the negative table must hold at least `num` entries)");
    if (!pair_row_ptr || !neg_table || !neg_out || !scratch || (batch > 0 && (!batch_nodes || !offsets || !node_out || !pos_out)))
        return ctgcn_set_error_(CTGCN_E_INVALID, "neg_sampling_indices: null pointer");
    if (batch > 0)
        hipLaunchKernelGGL(pos_sample_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, batch, batch_nodes,
                           pair_row_ptr, pair_col, (int)num, seed, offsets, node_out, pos_out);
    hipLaunchKernelGGL(neg_sample_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, table_len, neg_table, (int)num, seed ^ 0xabcdefull, neg_out, scratch);
    WK_TRY(hipGetLastError());
    return CTGCN_OK;
}
