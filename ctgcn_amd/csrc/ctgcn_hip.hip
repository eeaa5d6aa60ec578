// ctgcn_hip.hip — hand-written gfx950 (MI355X, CDNA4) kernels for the CTGCN hot path and the
// C ABI declared in include/ctgcn_hip.h.  Built with
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC
// No torch types, no CUDA shims, no CPU fallback.
//
// Kernels (all HBM/latency bound — integer or fp32 gather work, nothing here is GEMM shaped):
//   agg_fwd_kernel      CoreDiffusion aggregation loop (reference layers.py:41-48,58) fused over the
//                       K nested k-core matrices; also serves plain CSR SpMM (layers.py:43,45).
//   agg_bwd_kernel      its gradient w.r.t. X (slot-indexed gather of the suffix-summed dH).
//   agg_bwd_prep_kernel relu-mask + (double) suffix sum along the core axis.
//   kcore_*             level-synchronous k-core peel (structure_generation.py:35).
//   edge_level_kernel   level(e)=min(core[u],core[v]) + per-level histogram (structure_generation.py:48-53).
//   slot_reorder_kernel stable per-row partition by slot (helper.py:63-78 encoded as a table).
//
// Mapping used by the aggregation kernels (wave = 64 lanes):
//   a row of X is d fp32 = d/4 float4 "chunks".  A row of the sparse matrix is owned by a group of
//   LPR lanes (LPR = 8..64, power of two >= number of chunks, so d=128 -> 32 lanes, two rows per wave);
//   lane i of the group owns chunk i of every gathered X row, so one gather is ONE coalesced
//   16 B/lane load of the full 4*d-byte row.  The group's lanes first load LPR (col,val,slot) triples
//   with one coalesced load each and then broadcast them with ds_bpermute, U gathers are issued
//   back to back before the first is consumed.  Because the entries of a row are sorted by slot, the
//   K outputs need only two accumulators per lane (P = running A_j·x, R = running res_j) whatever K is.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/ctgcn_hip.h"
#include "ctgcn_jitter.h"
#include "ctgcn_table.h"          // diagnostic builds (-DCTGCN_JITTER): delays around every barrier; nothing in the product

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return fail(CTGCN_E_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));


template <int VEC> struct vec_of;
template <> struct vec_of<4> { using type = f4; };
template <> struct vec_of<1> { using type = float; };

template <int VEC> __device__ __forceinline__ typename vec_of<VEC>::type vzero();
template <> __device__ __forceinline__ f4 vzero<4>() { return f4{0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

__device__ __forceinline__ f4 vmax0(f4 v) { return f4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }
__device__ __forceinline__ float vmax0(float v) { return fmaxf(v, 0.f); }
__device__ __forceinline__ f4 vfma(float a, f4 x, f4 acc)
{
    return f4{fmaf(a, x.x, acc.x), fmaf(a, x.y, acc.y), fmaf(a, x.z, acc.z), fmaf(a, x.w, acc.w)};
}
__device__ __forceinline__ float vfma(float a, float x, float acc) { return fmaf(a, x, acc); }

struct AggArgs {
    int64_t n;
    int32_t d, K;
    const int32_t *row_ptr;
    const int32_t *col;
    const float *val;
    const uint8_t *slot;   // may be null: every entry is slot 0
    const float *src;      // X [n, ldsrc]  (fwd)   or Z [n, K, d] (bwd)
    int64_t ldsrc;
    const float *self;     // bwd only: S0 [n, d] or null
    float *out;            // fwd: H (row stride out_ld, slot stride d) ; bwd: dX (row stride out_ld)
    int64_t out_ld;
    uint32_t flags;
    int32_t accumulate;
    int32_t chunks;        // ceil(d / VEC)
    int32_t passes;        // ceil(chunks / LPR)
    const int32_t *long_rows;   // rows longer than long_thresh: skipped by the row-per-group kernels, one block each
    int32_t n_long;
    int32_t long_thresh;
    int32_t hub_groups;    // lane groups of the hub block that share one long row
    int32_t hub_compact;   // fwd hub kernel: write hub row i of long_rows to out row i (a compact scratch) instead of its own row
    // very long hub rows: hub_split > 1 blocks per hub row (grid = n_long * hub_split; a row of L entries uses ceil(L / HUB_SPLIT_ENTRIES) of
    // them, at most hub_split), each leaving the per-slot partial sums of its piece in hub_part [n_long][hub_split][K (fwd) | 1 (bwd)][hub_ld];
    // agg_*_hub_final_kernel adds the pieces in piece order (deterministic) and finishes the row.  hub_part null: one block per row.
    int32_t hub_split;
    int32_t hub_ld;        // floats per partial vector (the feature width rounded up to whole passes)
    float *hub_part;
    // agg_fwd_split*_kernel with the GRU layer kernel as consumer (d = 128): rows are processed in `order` (position p takes matrix row
    // order[p], its K output rows are rows p K .. p K + K - 1 of the planes) and slot j of position p is only written when bit j of
    // tmask[p / 16] is set — a row whose first entry is tagged f has H[row, 0] = ... = H[row, f - 1] (nothing but the self loop has arrived),
    // the consumer multiplies the repeated row by W_ih once.  Both null: natural order, every slot written.
    const int32_t *order;
    const uint32_t *tmask;
    // GEMM consumer (d != 128): tiles of 64 positions (tile_shift 6: the row tile of gru_seq_h2_kernel) and COMPACT operand rows — the
    // written slots of tile T start at row tbase[T], position p of the tile owns popcount(mask) consecutive rows from
    // tbase[T] + (p % 64) popcount(mask); the GEMM then runs over the compact rows only.  tbase null: rows p K + slot (holes).
    int32_t tile_shift;
    const int32_t *tbase;
};

// ------------------------------------------------------------------------------------------------
// forward: per row   R = [self] X[row];  for slot j: P (+)= sum_{e in slot j} val*X[col];  R += P;
//                    out[row, j] = relu?(R)
// ------------------------------------------------------------------------------------------------
template <int VEC, int LPR, int U>
__global__ __launch_bounds__(256) void agg_fwd_kernel(const AggArgs a)
{
    using V = typename vec_of<VEC>::type;
    const int lig = threadIdx.x & (LPR - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + (threadIdx.x / LPR);
    if (row >= a.n) return;
    const int start = a.row_ptr[row], end = a.row_ptr[row + 1];
    if (end - start > a.long_thresh) return;          // hub row: agg_fwd_hub_kernel
    const bool self = (a.flags & CTGCN_F_SELF_LOOP) != 0;
    const bool relu = (a.flags & CTGCN_F_RELU) != 0;
    const bool nested = (a.flags & CTGCN_F_NESTED) != 0;
    const uint8_t *__restrict__ slot = a.slot;
    const float *__restrict__ X = a.src;
    float *__restrict__ outrow = a.out + row * a.out_ld;

    for (int pass = 0; pass < a.passes; ++pass) {
        const int ch = pass * LPR + lig;
        const bool live = ch < a.chunks;
        // dead lanes read chunk 0 (valid memory) and never store: keeps every load unconditional
        const int64_t foff = live ? (int64_t)ch * VEC : 0;
        V R = vzero<VEC>(), P = vzero<VEC>();
        if (self) R = *(const V *)(X + row * a.ldsrc + foff);
        int cur = 0;

        auto close_slot = [&]() {
            R += P;
            if (!nested) P = vzero<VEC>();
            V v = relu ? vmax0(R) : R;
            if (live) {
                V *o = (V *)(outrow + (int64_t)cur * a.d + foff);
                if (a.accumulate) v += *o;
                __builtin_nontemporal_store(v, o);
            }
            ++cur;
        };

        for (int base = start; base < end; base += LPR) {
            const int my = base + lig;
            int c = 0, s = 0;
            float w = 0.f;
            if (my < end) {
                c = a.col[my];
                w = a.val[my];
                s = slot ? (int)slot[my] : 0;
            }
            const int cnt = min(LPR, end - base);
            int j = 0;
            for (; j + U <= cnt; j += U) {
                V xv[U];
                float wj[U];
                int sj[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int cj = __shfl(c, j + u, LPR);
                    wj[u] = __shfl(w, j + u, LPR);
                    sj[u] = __shfl(s, j + u, LPR);
                    xv[u] = *(const V *)(X + (int64_t)cj * a.ldsrc + foff);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    while (cur < sj[u]) close_slot();
                    P = vfma(wj[u], xv[u], P);
                }
            }
            for (; j < cnt; ++j) {
                const int cj = __shfl(c, j, LPR);
                const float w1 = __shfl(w, j, LPR);
                const int s1 = __shfl(s, j, LPR);
                const V x1 = *(const V *)(X + (int64_t)cj * a.ldsrc + foff);
                while (cur < s1) close_slot();
                P = vfma(w1, x1, P);
            }
        }
        while (cur < a.K) close_slot();
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dX[row] = [self] S0[row] + sum_e val[e] * Z[col[e], slot[e], :]
// ------------------------------------------------------------------------------------------------
template <int VEC, int LPR, int U>
__global__ __launch_bounds__(256) void agg_bwd_kernel(const AggArgs a)
{
    using V = typename vec_of<VEC>::type;
    const int lig = threadIdx.x & (LPR - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / LPR) + (threadIdx.x / LPR);
    if (row >= a.n) return;
    const int start = a.row_ptr[row], end = a.row_ptr[row + 1];
    if (end - start > a.long_thresh) return;          // hub row: agg_bwd_hub_kernel
    const uint8_t *__restrict__ slot = a.slot;
    const float *__restrict__ Z = a.src;
    const int64_t zrow = (int64_t)a.K * a.d;

    for (int pass = 0; pass < a.passes; ++pass) {
        const int ch = pass * LPR + lig;
        const bool live = ch < a.chunks;
        const int64_t foff = live ? (int64_t)ch * VEC : 0;
        V P = vzero<VEC>();
        if (a.self) P = *(const V *)(a.self + row * (int64_t)a.d + foff);
        for (int base = start; base < end; base += LPR) {
            const int my = base + lig;
            int64_t off = 0;
            float w = 0.f;
            if (my < end) {
                off = (int64_t)a.col[my] * zrow + (slot ? (int64_t)slot[my] * a.d : 0);
                w = a.val[my];
            }
            const int cnt = min(LPR, end - base);
            int j = 0;
            for (; j + U <= cnt; j += U) {
                V xv[U];
                float wj[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t oj = __shfl(off, j + u, LPR);
                    wj[u] = __shfl(w, j + u, LPR);
                    xv[u] = *(const V *)(Z + oj + foff);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) P = vfma(wj[u], xv[u], P);
            }
            for (; j < cnt; ++j) {
                const int64_t oj = __shfl(off, j, LPR);
                const float w1 = __shfl(w, j, LPR);
                P = vfma(w1, *(const V *)(Z + oj + foff), P);
            }
        }
        if (live) {
            V *o = (V *)(a.out + row * a.out_ld + foff);
            if (a.accumulate) P += *o;
            *o = P;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Hub rows (longer than long_thresh entries): one 1024-thread block per row.  The row's entry range is cut into
// `hub_groups` contiguous segments, one per lane group; every group accumulates per-slot partial sums of ITS segment
// into LDS, group 0 then adds the partials in group order (deterministic) and runs the same R/P recurrence.
// ------------------------------------------------------------------------------------------------
constexpr int HUB_THREADS = 1024;
constexpr int HUB_SPLIT_ENTRIES = 8192;                  // entries per block when a hub row is cut into pieces
__host__ __device__ __forceinline__ int hub_pieces(int len, int max_pieces)
{
    const int np = (len + HUB_SPLIT_ENTRIES - 1) / HUB_SPLIT_ENTRIES;
    return np < 1 ? 1 : (np > max_pieces ? max_pieces : np);
}

template <int VEC, int LPR, int U>
__global__ __launch_bounds__(HUB_THREADS) void agg_fwd_hub_kernel(const AggArgs a)
{
    using V = typename vec_of<VEC>::type;
    extern __shared__ __align__(16) unsigned char hub_smem[];
    V *part = reinterpret_cast<V *>(hub_smem);            // [G][K][LPR]
    const int lig = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR, G = a.hub_groups;
    const bool pieces = a.hub_part != nullptr;            // several blocks per row: this one leaves the partial sums of its piece
    const int hub = pieces ? (int)(blockIdx.x / a.hub_split) : (int)blockIdx.x, piece = pieces ? (int)(blockIdx.x % a.hub_split) : 0;
    const int64_t row = a.long_rows[hub];
    int start = a.row_ptr[row], end = a.row_ptr[row + 1];
    if (pieces) {
        const int np = hub_pieces(end - start, a.hub_split);
        if (piece >= np) return;
        const int plen = ((end - start + np - 1) / np + LPR - 1) / LPR * LPR;
        start = min(end, start + piece * plen);
        end = min(end, start + plen);
    }
    const int seg = ((end - start + G - 1) / G + LPR - 1) / LPR * LPR;
    const int my_s = min(end, start + grp * seg), my_e = min(end, my_s + seg);
    const bool self = (a.flags & CTGCN_F_SELF_LOOP) != 0, relu = (a.flags & CTGCN_F_RELU) != 0;
    const bool nested = (a.flags & CTGCN_F_NESTED) != 0;
    const uint8_t *__restrict__ slot = a.slot;
    const float *__restrict__ X = a.src;
    float *__restrict__ outrow = a.out + (a.hub_compact ? (int64_t)hub : row) * a.out_ld;

    for (int pass = 0; pass < a.passes; ++pass) {
        const int ch = pass * LPR + lig;
        const bool live = ch < a.chunks;
        const int64_t foff = live ? (int64_t)ch * VEC : 0;
        if (grp < G) {
            for (int k = 0; k < a.K; ++k) part[((int64_t)grp * a.K + k) * LPR + lig] = vzero<VEC>();
            V P = vzero<VEC>();
            int cur = -1;
            for (int base = my_s; base < my_e; base += LPR) {
                const int my = base + lig;
                int c = 0, s = 0;
                float w = 0.f;
                if (my < my_e) { c = a.col[my]; w = a.val[my]; s = slot ? (int)slot[my] : 0; }
                const int cnt = min(LPR, my_e - base);
                int j = 0;
                for (; j + U <= cnt; j += U) {
                    V xv[U];
                    float wj[U];
                    int sj[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int cj = __shfl(c, j + u, LPR);
                        wj[u] = __shfl(w, j + u, LPR);
                        sj[u] = __shfl(s, j + u, LPR);
                        xv[u] = *(const V *)(X + (int64_t)cj * a.ldsrc + foff);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (sj[u] != cur) {
                            if (cur >= 0) part[((int64_t)grp * a.K + cur) * LPR + lig] = P;
                            P = vzero<VEC>();
                            cur = sj[u];
                        }
                        P = vfma(wj[u], xv[u], P);
                    }
                }
                for (; j < cnt; ++j) {
                    const int cj = __shfl(c, j, LPR);
                    const float w1 = __shfl(w, j, LPR);
                    const int s1 = __shfl(s, j, LPR);
                    const V x1 = *(const V *)(X + (int64_t)cj * a.ldsrc + foff);
                    if (s1 != cur) {
                        if (cur >= 0) part[((int64_t)grp * a.K + cur) * LPR + lig] = P;
                        P = vzero<VEC>();
                        cur = s1;
                    }
                    P = vfma(w1, x1, P);
                }
            }
            if (cur >= 0) part[((int64_t)grp * a.K + cur) * LPR + lig] = P;
        }
        __syncthreads();
        if (pieces) {
            if (grp == 0 && live) {
                for (int k = 0; k < a.K; ++k) {
                    V sk = vzero<VEC>();
                    for (int g = 0; g < G; ++g) sk += part[((int64_t)g * a.K + k) * LPR + lig];
                    *(V *)(a.hub_part + (((int64_t)hub * a.hub_split + piece) * a.K + k) * a.hub_ld + foff) = sk;
                }
            }
            __syncthreads();
            continue;
        }
        if (grp == 0) {
            V R = vzero<VEC>(), Pc = vzero<VEC>();
            if (self) R = *(const V *)(X + row * a.ldsrc + foff);
            for (int k = 0; k < a.K; ++k) {
                V sk = vzero<VEC>();
                for (int g = 0; g < G; ++g) sk += part[((int64_t)g * a.K + k) * LPR + lig];
                Pc = nested ? Pc + sk : sk;
                R += Pc;
                V v = relu ? vmax0(R) : R;
                if (live) {
                    V *o = (V *)(outrow + (int64_t)k * a.d + foff);
                    if (a.accumulate) v += *o;
                    *o = v;
                }
            }
        }
        __syncthreads();
    }
}

template <int VEC, int LPR, int U>
__global__ __launch_bounds__(HUB_THREADS) void agg_bwd_hub_kernel(const AggArgs a)
{
    using V = typename vec_of<VEC>::type;
    extern __shared__ __align__(16) unsigned char hub_smem[];
    V *part = reinterpret_cast<V *>(hub_smem);            // [G][LPR]
    const int lig = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR, G = a.hub_groups;
    const bool pieces = a.hub_part != nullptr;
    const int hub = pieces ? (int)(blockIdx.x / a.hub_split) : (int)blockIdx.x, piece = pieces ? (int)(blockIdx.x % a.hub_split) : 0;
    const int64_t row = a.long_rows[hub];
    int start = a.row_ptr[row], end = a.row_ptr[row + 1];
    if (pieces) {
        const int np = hub_pieces(end - start, a.hub_split);
        if (piece >= np) return;
        const int plen = ((end - start + np - 1) / np + LPR - 1) / LPR * LPR;
        start = min(end, start + piece * plen);
        end = min(end, start + plen);
    }
    const int seg = ((end - start + G - 1) / G + LPR - 1) / LPR * LPR;
    const int my_s = min(end, start + grp * seg), my_e = min(end, my_s + seg);
    const uint8_t *__restrict__ slot = a.slot;
    const float *__restrict__ Z = a.src;
    const int64_t zrow = (int64_t)a.K * a.d;

    for (int pass = 0; pass < a.passes; ++pass) {
        const int ch = pass * LPR + lig;
        const bool live = ch < a.chunks;
        const int64_t foff = live ? (int64_t)ch * VEC : 0;
        if (grp < G) {
            V P = vzero<VEC>();
            for (int base = my_s; base < my_e; base += LPR) {
                const int my = base + lig;
                int64_t off = 0;
                float w = 0.f;
                if (my < my_e) {
                    off = (int64_t)a.col[my] * zrow + (slot ? (int64_t)slot[my] * a.d : 0);
                    w = a.val[my];
                }
                const int cnt = min(LPR, my_e - base);
                int j = 0;
                for (; j + U <= cnt; j += U) {
                    V xv[U];
                    float wj[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int64_t oj = __shfl(off, j + u, LPR);
                        wj[u] = __shfl(w, j + u, LPR);
                        xv[u] = *(const V *)(Z + oj + foff);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) P = vfma(wj[u], xv[u], P);
                }
                for (; j < cnt; ++j) {
                    const int64_t oj = __shfl(off, j, LPR);
                    const float w1 = __shfl(w, j, LPR);
                    P = vfma(w1, *(const V *)(Z + oj + foff), P);
                }
            }
            part[(int64_t)grp * LPR + lig] = P;
        }
        __syncthreads();
        if (pieces) {
            if (grp == 0 && live) {
                V P = vzero<VEC>();
                for (int g = 0; g < G; ++g) P += part[(int64_t)g * LPR + lig];
                *(V *)(a.hub_part + ((int64_t)hub * a.hub_split + piece) * a.hub_ld + foff) = P;
            }
            __syncthreads();
            continue;
        }
        if (grp == 0) {
            V P = vzero<VEC>();
            if (a.self) P = *(const V *)(a.self + row * (int64_t)a.d + foff);
            for (int g = 0; g < G; ++g) P += part[(int64_t)g * LPR + lig];
            if (live) {
                V *o = (V *)(a.out + row * a.out_ld + foff);
                if (a.accumulate) P += *o;
                *o = P;
            }
        }
        __syncthreads();
    }
}

// Second pass for hub rows cut into pieces: one small block per hub row adds the pieces' partial sums in piece order and runs the
// row's R/P recurrence (forward) / adds the self term (backward).  Deterministic: fixed piece boundaries, fixed order.
template <int VEC, bool FWD>
__global__ __launch_bounds__(256) void agg_hub_final_kernel(const AggArgs a)
{
    using V = typename vec_of<VEC>::type;
    const int hub = blockIdx.x;
    const int64_t row = a.long_rows[hub];
    const int np = hub_pieces(a.row_ptr[row + 1] - a.row_ptr[row], a.hub_split);
    const bool self = (a.flags & CTGCN_F_SELF_LOOP) != 0, relu = (a.flags & CTGCN_F_RELU) != 0, nested = (a.flags & CTGCN_F_NESTED) != 0;
    const int KK = FWD ? a.K : 1;
    const float *__restrict__ mine = a.hub_part + (int64_t)hub * a.hub_split * KK * a.hub_ld;
    for (int ch = threadIdx.x; ch < a.chunks; ch += 256) {
        const int64_t foff = (int64_t)ch * VEC;
        if (FWD) {
            float *__restrict__ outrow = a.out + (a.hub_compact ? (int64_t)hub : row) * a.out_ld;
            V R = vzero<VEC>(), Pc = vzero<VEC>();
            if (self) R = *(const V *)(a.src + row * a.ldsrc + foff);
            for (int k = 0; k < a.K; ++k) {
                V sk = vzero<VEC>();
                for (int p = 0; p < np; ++p) sk += *(const V *)(mine + ((int64_t)p * a.K + k) * a.hub_ld + foff);
                Pc = nested ? Pc + sk : sk;
                R += Pc;
                V v = relu ? vmax0(R) : R;
                V *o = (V *)(outrow + (int64_t)k * a.d + foff);
                if (a.accumulate) v += *o;
                *o = v;
            }
        } else {
            V P = vzero<VEC>();
            if (a.self) P = *(const V *)(a.self + row * (int64_t)a.d + foff);
            for (int p = 0; p < np; ++p) P += *(const V *)(mine + (int64_t)p * a.hub_ld + foff);
            V *o = (V *)(a.out + row * a.out_ld + foff);
            if (a.accumulate) P += *o;
            *o = P;
        }
    }
}

// G_j = dH_j*[H_j>0]; S_j = sum_{i>=j} G_i; Z_j = nested ? sum_{i>=j} S_i : S_j; S0 = S_0
template <int VEC>
__global__ __launch_bounds__(256) void agg_bwd_prep_kernel(int64_t n, int32_t d, int32_t K, int32_t chunks,
                                                           const float *__restrict__ dH, const float *__restrict__ H,
                                                           float *__restrict__ Z, float *__restrict__ S0, uint32_t flags)
{
    using V = typename vec_of<VEC>::type;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * chunks) return;
    const int64_t row = idx / chunks;
    const int64_t foff = (idx - row * chunks) * VEC;
    const bool relu = (flags & CTGCN_F_RELU) != 0, nested = (flags & CTGCN_F_NESTED) != 0;
    V s = vzero<VEC>(), t = vzero<VEC>();
    for (int j = K - 1; j >= 0; --j) {
        const int64_t o = (row * K + j) * d + foff;
        V g = *(const V *)(dH + o);
        if (relu) {
            const V h = *(const V *)(H + o);
            if constexpr (VEC == 4) {
                g.x = h.x > 0.f ? g.x : 0.f; g.y = h.y > 0.f ? g.y : 0.f;
                g.z = h.z > 0.f ? g.z : 0.f; g.w = h.w > 0.f ? g.w : 0.f;
            } else {
                g = h > 0.f ? g : 0.f;
            }
        }
        s += g;
        t += s;
        *(V *)(Z + o) = nested ? t : s;
    }
    if (S0) *(V *)(S0 + row * d + foff) = s;
}

// ---------------------------------------------------------------------------------- launch helpers
struct AggPlan { int vec, lpr, chunks, passes; };

AggPlan plan_for(int d, bool vec4_ok)
{
    AggPlan p;
    p.vec = vec4_ok ? 4 : 1;
    p.chunks = (d + p.vec - 1) / p.vec;
    int lpr = 8;
    while (lpr < 64 && lpr < p.chunks) lpr <<= 1;
    p.lpr = lpr;
    p.passes = (p.chunks + lpr - 1) / lpr;
    return p;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// CUs a persistent kernel (one block per CU) may count on: the device's, or fewer when the caller runs it on a stream with a CU mask
// (ctgcn_set_persistent_cus: the snapshot pipeline gives the HBM-bound aggregation its own few CUs next to the matrix-core kernels)
// (round 4's gru_layer8_h2_pair_kernel — two tiles in flight per block, the two waves of a SIMD half a step out of phase — was bit-identical and
// 1.9 - 2.6 x slower, profiles/r04_layer_pair_ab.txt; removed in round 5, git history has it.)
#ifdef CTGCN_LAYER_TIMELINE
// diagnostic build: per (block, wave) phase sums of the layer kernel, written to $CTGCN_LAYER_TIMELINE_FILE by every call (tools/layer_timeline.py)
struct LayerArgs;
const char *timeline_begin(LayerArgs &a, unsigned nb8, void *stream);
void timeline_end(LayerArgs &a, unsigned nb8, const char *tl_file, void *stream);
#endif
int g_persistent_cus = 0;
int persistent_cus(int device_cus) { return g_persistent_cus > 0 && g_persistent_cus < device_cus ? g_persistent_cus : device_cus; }


constexpr size_t HUB_LDS_BUDGET = 144 * 1024;

template <bool FWD, int VEC, int LPR>
void launch_agg_t(AggArgs a, hipStream_t st)
{
    constexpr int U = 4;
    const int rows_per_block = 256 / LPR;
    const int64_t blocks = (a.n + rows_per_block - 1) / rows_per_block;
    if (FWD)
        hipLaunchKernelGGL((agg_fwd_kernel<VEC, LPR, U>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((agg_bwd_kernel<VEC, LPR, U>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    if (a.n_long > 0) {
        const size_t per_group = (size_t)(FWD ? a.K : 1) * LPR * sizeof(typename vec_of<VEC>::type);
        int G = HUB_THREADS / LPR;
        while (G > 1 && per_group * G > HUB_LDS_BUDGET) --G;
        a.hub_groups = G;
        const size_t lds = per_group * G;
        const unsigned hub_grid = (unsigned)a.n_long * (unsigned)(a.hub_part ? a.hub_split : 1);
        if (FWD) {
            auto k = agg_fwd_hub_kernel<VEC, LPR, U>;
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(k, dim3(hub_grid), dim3(HUB_THREADS), lds, st, a);
        } else {
            auto k = agg_bwd_hub_kernel<VEC, LPR, U>;
            hipLaunchKernelGGL(k, dim3(hub_grid), dim3(HUB_THREADS), lds, st, a);
        }
        if (a.hub_part) hipLaunchKernelGGL((agg_hub_final_kernel<VEC, FWD>), dim3((unsigned)a.n_long), dim3(256), 0, st, a);
    }
}

// hub rows in pieces (AggArgs::hub_split / hub_part): validated against the caller's workspace; slots = K (forward) or 1 (backward)
size_t hub_workspace_bytes_(int64_t n_long, int32_t hub_split, int32_t slots, int32_t d)
{
    if (n_long <= 0 || hub_split <= 1) return 0;
    return (size_t)n_long * hub_split * slots * ((d + 3) / 4 * 4) * sizeof(float);
}
int set_hub_pieces(AggArgs &a, int32_t slots, int32_t hub_split, void *ws, size_t ws_bytes, const char *who)
{
    a.hub_split = 1; a.hub_part = nullptr; a.hub_ld = (a.d + 3) / 4 * 4;
    if (a.n_long <= 0 || hub_split <= 1 || !ws) return CTGCN_OK;
    if (hub_split > 64) return fail(CTGCN_E_INVALID, "%s: hub_split=%d outside [1,64]", who, hub_split);
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) || ws_bytes < hub_workspace_bytes_(a.n_long, hub_split, slots, a.d))
        return fail(CTGCN_E_WORKSPACE, "%s: hub workspace must be 16-byte aligned and hold ctgcn_hub_workspace_bytes() bytes", who);
    a.hub_split = hub_split; a.hub_part = (float *)ws;
    return CTGCN_OK;
}

template <bool FWD>
int launch_agg(AggArgs a, bool vec4_ok, hipStream_t st)
{
    const AggPlan p = plan_for(a.d, vec4_ok);
    a.chunks = p.chunks;
    a.passes = p.passes;
    if (a.n == 0) return CTGCN_OK;
    if (a.n_long <= 0 || !a.long_rows) { a.n_long = 0; a.long_thresh = 0x7fffffff; }
    if (a.n_long > 0 && (size_t)a.K * p.lpr * (p.vec * 4) > HUB_LDS_BUDGET) { a.n_long = 0; a.long_thresh = 0x7fffffff; }   // K*d too large for the LDS partials: rows stay on the normal path
    const int64_t rows_per_block = 256 / p.lpr;
    if ((a.n + rows_per_block - 1) / rows_per_block > 0x7fffffffLL) return fail(CTGCN_E_UNSUPPORTED, "grid too large");
#define CASE(V, L) if (p.vec == V && p.lpr == L) { launch_agg_t<FWD, V, L>(a, st); }
    CASE(4, 8) else CASE(4, 16) else CASE(4, 32) else CASE(4, 64)
    else CASE(1, 8) else CASE(1, 16) else CASE(1, 32) else CASE(1, 64)
#undef CASE
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

// ================================================================================================
// k-core peel.  One kernel launch per level k (the launch boundary is the only grid-wide barrier):
//   scan  — each block scans its own vertex range for unclaimed vertices with deg == k;
//   chase — the block processes its queue; a neighbour whose degree drops to k is claimed by the
//           thread whose atomicSub returned k+1 and appended to THIS block's queue (LDS, spilling
//           to a per-block linked stack inside a shared pool of n entries: every vertex is pushed
//           at most once in the whole run, so the pool cannot overflow).
// deg[] only changes through device-scope atomics; a vertex's final deg is its core number.
// ================================================================================================
struct KcoreCtl {
    int visited;
    int pool_tail;
    int max_core;
    int pad[13];
};

constexpr int KC_QCAP = 8192;
constexpr int KC_GROUP = 16;   // lanes cooperating on one vertex's neighbour list ...
constexpr int KC_LONG = 48;    // ... unless it is longer than this: then a whole wave takes it

__global__ __launch_bounds__(256) void kcore_init_kernel(int n, const int32_t *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ col, int32_t *__restrict__ deg)
{
    const int lig = threadIdx.x & 7;
    const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    if (v >= n) return;
    int cnt = 0;
    for (int e = row_ptr[v] + lig, end = row_ptr[v + 1]; e < end; e += 8) cnt += (col[e] != (int)v);
    cnt += __shfl_xor(cnt, 1, 8);
    cnt += __shfl_xor(cnt, 2, 8);
    cnt += __shfl_xor(cnt, 4, 8);
    if (lig == 0) deg[v] = cnt;
}

__device__ __forceinline__ bool kc_claim(unsigned *claimed, int v)
{
    const unsigned bit = 1u << (v & 31);
    return (atomicOr(&claimed[v >> 5], bit) & bit) == 0;
}

__global__ __launch_bounds__(256) void kcore_level_kernel(int n, int k, int chunk, const int32_t *__restrict__ row_ptr,
                                                          const int32_t *__restrict__ col, int32_t *deg,
                                                          unsigned *claimed, int32_t *pool_v, int32_t *pool_prev,
                                                          KcoreCtl *ctl)
{
    __shared__ int q[KC_QCAP];
    __shared__ int s_tail, s_top;
    const int tid = threadIdx.x;
    if (tid == 0) { s_tail = 0; s_top = -1; }
    __syncthreads();

    auto push = [&](int u) {
        const int pos = atomicAdd(&s_tail, 1);
        if (pos < KC_QCAP) {
            q[pos] = u;
        } else {
            const int p = atomicAdd(&ctl->pool_tail, 1);
            pool_v[p] = u;
            pool_prev[p] = atomicExch(&s_top, p);
        }
    };

    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (int v = lo + tid; v < hi; v += 256)
        if (__hip_atomic_load(&deg[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k && kc_claim(claimed, v)) push(v);

    const int grp = tid / KC_GROUP, lig = tid % KC_GROUP, ngrp = 256 / KC_GROUP;
    const int wv = tid >> 6, wl = tid & 63;
    // remove v: decrement every live neighbour; whoever takes a neighbour from k+1 to k owns (claims + queues) it
    auto relax = [&](int v, int u) {
        if (u == v) return;
        if (__hip_atomic_load(&deg[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > k) {
            const int old = atomicSub(&deg[u], 1);
            if (old == k + 1) {
                if (kc_claim(claimed, u)) push(u);
            } else if (old <= k) {
                atomicAdd(&deg[u], 1);
            }
        }
    };
    int begin = 0, processed = 0;
    for (;;) {
        __syncthreads();                   // the pushes of the round before are done
        int tail = min(s_tail, KC_QCAP);
        // EVERY thread has read s_tail before anybody pushes again.  (Rounds 1-4 had this barrier in the refill branch only: a wave that
        // read s_tail late — after a faster wave's first push of the new round — saw another `tail`, took another side of the branch below
        // and met the others at different barriers.  Pushes sit behind two dependent global loads, so it never showed; the barrier-jitter
        // build of round 5, ctgcn_jitter.h, produced wrong core numbers within seconds.)
        __syncthreads();
        if (begin == tail) {
            if (tid == 0) {                // refill from this block's spill stack (rare)
                int m = 0, top = s_top;
                while (top >= 0 && m < KC_QCAP / 2) { q[m++] = pool_v[top]; top = pool_prev[top]; }
                s_top = top;
                s_tail = m;
            }
            __syncthreads();
            begin = 0;
            tail = s_tail;
            __syncthreads();               // as above: read by everyone before the first push
            if (tail == 0) break;
        }
        // pass A: short neighbour lists, one 16-lane group per vertex; pass B: long lists (hubs, the dense top
        // cores), one whole wave per vertex — the chase is latency bound, so hubs must not crawl 16 lanes at a time
        for (int i = begin + grp; i < tail; i += ngrp) {
            const int v = q[i];
            const int s0 = row_ptr[v], e0 = row_ptr[v + 1];
            if (e0 - s0 > KC_LONG) continue;
            for (int e = s0 + lig; e < e0; e += KC_GROUP) relax(v, col[e]);
        }
        for (int i = begin + wv; i < tail; i += 4) {
            const int v = q[i];
            const int s0 = row_ptr[v], e0 = row_ptr[v + 1];
            if (e0 - s0 <= KC_LONG) continue;
            for (int e = s0 + wl; e < e0; e += 64) relax(v, col[e]);
        }
        processed += tail - begin;
        begin = tail;
    }
    if (tid == 0 && processed) {
        atomicAdd(&ctl->visited, processed);
        atomicMax(&ctl->max_core, k);
    }
}

// ------------------------------------------------------------------ k-core by local h-index sweeps (round 5)
// core(v) is the greatest fixed point below deg of  h(v) <- H({h(u) : u in N(v)}),  H = the largest k with at least k values >= k
// (Lu, Zhou, Zhang, Stanley 2016; the peel above needs one dependent cascade per level — 84 launches of ~86 us on the config-5 snapshot, a
// latency chain — where a sweep is one bandwidth-bound pass over the CSR and a few tens of sweeps converge on power-law graphs).  Values only
// ever decrease and never pass below the core number, in any update order, so sweeps update h IN PLACE and may read stale neighbours (another
// XCD's L2): a stale value is a larger one, the result stays an upper bound.  What must not be lost is the knowledge that a neighbour changed:
//   * a vertex is recomputed in sweep s when its byte in flags[s & 1] is set (the first FULL sweeps recompute everybody);
//   * a vertex v whose value drops from c to `now` stops counting towards exactly the neighbours u with now < h(u) <= c (it never counted for
//     h(u) > c, it still counts for h(u) <= now): it sets THEIR bytes in flags[(s + 1) & 1], and its own (plain byte stores of the value 1 —
//     concurrent writers agree; L2 lines carry byte masks, so bytes written under different XCDs merge at write-back); the block that owns a
//     flag clears it after reading it — two sweeps before anybody sets it again, with kernel boundaries in between.  Round 5 flagged every
//     neighbour with h(u) > now: a degree-5 vertex moving from 5 to 4 had its hub neighbour (h = 84, ~2 000 entries) recomputed from scratch,
//     and the ~1 000 hubs of the config-5 snapshot were recomputed in nearly every sweep (35 - 54 us of each).  Why the own flag: v reads
//     h(u) while u may be moving in the same sweep (or sits stale in another XCD's L2) — the value read is then LARGER than the true one and
//     may fail `<= c` although the true one passes.  But then u changed in this sweep, flagged itself, and is recomputed in the next one
//     from values that are all visible by then; an u that did not change is read exactly.  By induction no lost update;
//   * a sweep that sets no flag changed no vertex: the fixed point.  ctl->marks[s & 15] == 0; the NEXT sweep's kernels see that, set ctl->done
//     and return, and so does every kernel queued behind them: the host queues sweeps in one batch and reads the control block once.
// Same unique integers as the peel (tests compare both with the Batagelj-Zaversnik oracle); level_cap clips every value at the cap
// (H of clipped values, clipped, is the clipped H: counts of values >= k for k <= cap do not change).
constexpr int KH_CHUNK = 1024;      // vertices per block
constexpr int KH_CACHE = 1024;      // neighbour values a wave keeps in LDS for a long list

struct KhCtl {               // 256 bytes: the head of the workspace (the peel's KcoreCtl region)
    int marks[16];           // flags set by sweep s & 15
    int hubs[16];            // active hub vertices (lists longer than KH_CACHE) queued by sweep s & 15 for kcore_hindex_hub_kernel
    int max_core;
    int done;                // set on the device by the first sweep that finds its predecessor's counter at zero: every later kernel of the call returns at once
    int pad[30];
};
static_assert(sizeof(KhCtl) == 256, "KhCtl must fit the 256-byte control block of the k-core workspace");
constexpr int KH_BINS = 8192;

__device__ __forceinline__ bool kh_finished(KhCtl *ctl, int check_prev, int sweep)
{
    // uniform over the grid: `done` is only ever written by kernels that return here, marks[(sweep - 1) & 15] is final (kernel boundary)
    if (__hip_atomic_load(&ctl->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    if (check_prev && __hip_atomic_load(&ctl->marks[(sweep - 1) & 15], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(&ctl->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    return false;
}

__global__ __launch_bounds__(256) void kcore_hindex_kernel(int n, int cap, int all_active, int do_mark, int check_prev, int sweep, const int32_t *__restrict__ row_ptr,
                                                           const int32_t *__restrict__ col, int32_t *h, uint8_t *flag_in, uint8_t *flag_out, KhCtl *ctl,
                                                           int32_t *hub_list)
{
    if (kh_finished(ctl, check_prev, sweep)) return;
    // three queues by list length, filled by the threads that read the flags (each reads its vertices' row_ptr once, all in parallel: a
    // queue scanned by every phase with a row_ptr load per entry to skip the others' vertices cost ~200 us per sweep, active or not)
    __shared__ int q[3][KH_CHUNK];
    __shared__ int cache[4][KH_CACHE];
    __shared__ int s_tail[3], s_marks;
    constexpr int KH_TINY = 16;
    const int tid = threadIdx.x;
    if (tid < 3) s_tail[tid] = 0;
    if (tid == 0) s_marks = 0;
    __syncthreads();
    const int lo = blockIdx.x * KH_CHUNK;
    {   // this block's 1 024 flags, four per thread; read, clear, queue
        const int v0 = lo + tid * 4;
        uint32_t f = 0;
        if (v0 + 3 < n) {
            f = *(const uint32_t *)(flag_in + v0);
            if (f) *(uint32_t *)(flag_in + v0) = 0;
        } else {
            for (int i = 0; i < 4; ++i) if (v0 + i < n && flag_in[v0 + i]) { f |= 0xffu << (8 * i); flag_in[v0 + i] = 0; }
        }
        if (all_active) f = 0xffffffffu;
        if (f) {
            int rp[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) rp[i] = row_ptr[min(v0 + i, n)];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (((f >> (8 * i)) & 0xff) && v0 + i < n) {
                    const int deg = rp[i + 1] - rp[i];
                    if (deg == 0) continue;               // h = 0 from the start
                    if (deg > KH_CACHE) { hub_list[atomicAdd(&ctl->hubs[sweep & 15], 1)] = v0 + i; continue; }      // a block per hub: next kernel
                    const int which = deg <= KH_TINY ? 0 : (deg <= KC_LONG ? 1 : 2);
                    q[which][atomicAdd(&s_tail[which], 1)] = v0 + i;
                }
        }
    }
    __syncthreads();
    int marks = 0;
    // tiny lists (<= 16 entries: most vertices of a power-law graph): one LANE per vertex.  All 16 column loads, then all 16 value gathers, are
    // independent requests — a 16-lane group per vertex walked its 64 vertices one dependent chain (row_ptr -> col -> h) after the other and a
    // sweep took ~200 us whatever the number of active vertices.  The values (clipped at c <= 16) are counted in sixteen 8-bit counters.
    for (int i = tid, tail = s_tail[0]; i < tail; i += 256) {
        const int v = q[0][i];
        const int s0 = row_ptr[v], deg = row_ptr[v + 1] - s0;
        const int c = h[v];
        if (c == 0) continue;
        int nb[KH_TINY], hv[KH_TINY];
#pragma unroll
        for (int j = 0; j < KH_TINY; ++j) nb[j] = j < deg ? col[s0 + j] : v;
        unsigned le = 0;                                  // bit j: the neighbour's value is <= c (v counted towards it before this update)
#pragma unroll
        for (int j = 0; j < KH_TINY; ++j) {
            const int raw = nb[j] != v ? h[nb[j]] : 0;
            le |= (unsigned)(raw <= c) << j;
            hv[j] = min(raw, c);
        }
        unsigned long long c_lo = 0, c_hi = 0;            // counters of the values 1 .. 8 and 9 .. 16
#pragma unroll
        for (int j = 0; j < KH_TINY; ++j) {
            const int x = hv[j];
            if (x >= 9) c_hi += 1ull << (8 * (x - 9));
            else if (x >= 1) c_lo += 1ull << (8 * (x - 1));
        }
        int now = 0, cnt = 0;
        for (int k = c; k >= 1; --k) {
            cnt += (int)(((k >= 9 ? c_hi >> (8 * (k - 9)) : c_lo >> (8 * (k - 1)))) & 0xff);
            if (cnt >= k) { now = k; break; }
        }
        if (now < c) {
            h[v] = now;
            if (do_mark) {
                flag_out[v] = 1; ++marks;
#pragma unroll
                for (int j = 0; j < KH_TINY; ++j)
                    if (hv[j] > now && ((le >> j) & 1)) { flag_out[nb[j]] = 1; ++marks; }
            }
        }
    }
    // short lists (17 .. KC_LONG entries): one 16-lane group per vertex, up to KC_LONG / 16 values per lane
    const int grp = tid / KC_GROUP, lig = tid % KC_GROUP, ngrp = 256 / KC_GROUP;
    constexpr int PER = KC_LONG / KC_GROUP;
    for (int i = grp, tail = s_tail[1]; i < tail; i += ngrp) {
        const int v = q[1][i];
        const int s0 = row_ptr[v], e0 = row_ptr[v + 1];
        const int c = h[v];
        int val[PER], nb[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int e = s0 + lig + KC_GROUP * j;
            nb[j] = e < e0 ? col[e] : v;
            val[j] = nb[j] != v ? min(h[nb[j]], c + 1) : 0;       // c + 1 = "above c": counts like c for every k <= c, and tells the marking apart
        }
        int klo = 0, khi = c;
        while (klo < khi) {                               // uniform across the group: c and the counts are
            const int mid = (klo + khi + 1) >> 1;
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) cnt += val[j] >= mid;
            cnt += __shfl_xor(cnt, 1, KC_GROUP);
            cnt += __shfl_xor(cnt, 2, KC_GROUP);
            cnt += __shfl_xor(cnt, 4, KC_GROUP);
            cnt += __shfl_xor(cnt, 8, KC_GROUP);
            if (cnt >= mid) klo = mid; else khi = mid - 1;
        }
        if (klo < c) {
            if (lig == 0) h[v] = klo;
            if (do_mark) {
                if (lig == 0) { flag_out[v] = 1; ++marks; }
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (val[j] > klo && val[j] <= c) { flag_out[nb[j]] = 1; ++marks; }
            }
        }
    }
    // long lists (up to KH_CACHE entries): one wave per vertex, the clipped neighbour values in LDS; longer ones: kcore_hindex_hub_kernel
    const int wv = tid >> 6, wl = tid & 63;
    for (int i = wv, tail = s_tail[2]; i < tail; i += 4) {
        const int v = q[2][i];
        const int s0 = row_ptr[v], e0 = row_ptr[v + 1], deg = e0 - s0;
        const int c = h[v];
        // ALL of the list's (column, value) request pairs in flight at once — KH_CACHE / 64 = 16 per lane, the columns kept for the marking
        // pass: one pair per trip was a dependent chain of up to 16 trips x two memory latencies, and the marking pass loaded the columns
        // again one trip at a time (a 1 000-entry list in one block set the ~30 us floor of every tail sweep)
        constexpr int KH_PER = KH_CACHE / 64;
        int un[KH_PER];
#pragma unroll
        for (int j = 0; j < KH_PER; ++j) un[j] = wl + 64 * j < deg ? col[s0 + wl + 64 * j] : v;
#pragma unroll
        for (int j = 0; j < KH_PER; ++j) {
            const int x = un[j] != v ? min(h[un[j]], c + 1) : 0;
            if (wl + 64 * j < deg) cache[wv][wl + 64 * j] = x;
        }
        int klo = 0, khi = c;
        while (klo < khi) {
            const int mid = (klo + khi + 1) >> 1;
            int cnt = 0;
            for (int e = wl; e < deg; e += 64) cnt += cache[wv][e] >= mid;
#pragma unroll
            for (int o = 32; o; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (cnt >= mid) klo = mid; else khi = mid - 1;
        }
        if (klo < c) {
            if (wl == 0) h[v] = klo;
            if (do_mark) {
                if (wl == 0) { flag_out[v] = 1; ++marks; }
#pragma unroll
                for (int j = 0; j < KH_PER; ++j)
                    if (wl + 64 * j < deg) {
                        const int x = cache[wv][wl + 64 * j];
                        if (x > klo && x <= c) { flag_out[un[j]] = 1; ++marks; }
                    }
            }
        }
    }
    if (marks) atomicAdd(&s_marks, marks);
    __syncthreads();
    if (tid == 0 && s_marks) atomicAdd(&ctl->marks[sweep & 15], s_marks);
    (void)cap;
}

// hub vertices (lists longer than KH_CACHE) of one sweep, a block each: ONE gather pass into an LDS histogram of the clipped values, then the
// largest k with count(values >= k) >= k from per-thread suffix sums (a wave probing a 200 000-entry list eleven times would set the sweep's
// time).  Values are clipped at KH_BINS: a hub whose estimate is still above that comes out at KH_BINS at most — an upper bound, a decrease —
// and flags ITSELF for the next sweep.
__global__ __launch_bounds__(256) void kcore_hindex_hub_kernel(int do_mark, int check_prev, int sweep, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col, int32_t *h,
                                                               uint8_t *flag_out, KhCtl *ctl, const int32_t *__restrict__ hub_list)
{
    __shared__ int hist[KH_BINS + 1];
    __shared__ int part[256];
    __shared__ int s_best, s_marks;
    const int tid = threadIdx.x;
    if (kh_finished(ctl, check_prev, sweep)) return;
    // the counters of the NEXT sweep (its kernels start after this one has finished; nobody else touches that slot now)
    if (blockIdx.x == 0 && tid == 0) { ctl->marks[(sweep + 1) & 15] = 0; ctl->hubs[(sweep + 1) & 15] = 0; }
    const int count = ctl->hubs[sweep & 15];
    int marks = 0;
    for (int i = blockIdx.x; i < count; i += gridDim.x) {
        const int v = hub_list[i];
        const int s0 = row_ptr[v], e0 = row_ptr[v + 1];
        const int c = h[v];
        const int B = min(c, KH_BINS);
        __syncthreads();
        for (int b = tid; b <= B; b += 256) hist[b] = 0;
        if (tid == 0) { s_best = 0; s_marks = 0; }
        __syncthreads();
        for (int e = s0 + tid; e < e0; e += 256 * 8) {   // eight independent (column, value) request pairs per thread in flight
            int u[8], x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) u[j] = e + 256 * j < e0 ? col[e + 256 * j] : v;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = u[j] != v ? min(h[u[j]], B) : 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (x[j] >= 1) atomicAdd(&hist[x[j]], 1);
        }
        __syncthreads();
        // thread t owns bins [t W + 1, (t + 1) W]: chunk sums, suffix over the chunks, then the bins from the top
        const int W = (B + 255) / 256;
        const int b0 = tid * W + 1, b1 = min(B, (tid + 1) * W);
        int sum = 0;
        for (int b = b0; b <= b1; ++b) sum += hist[b];
        part[tid] = sum;
        __syncthreads();
        int above = 0;
        for (int t = tid + 1; t < 256; ++t) above += part[t];
        int run = above, best = 0;
        for (int b = b1; b >= b0; --b) {
            run += hist[b];
            if (run >= b) { best = b; break; }
        }
        if (best) atomicMax(&s_best, best);
        __syncthreads();
        const int now = s_best;
        if (now < c) {
            if (tid == 0) {
                h[v] = now;
                if (do_mark || (now == B && B < c)) { flag_out[v] = 1; ++marks; }      // changed: look again (clipped at KH_BINS: not the h-index yet)
            }
            if (do_mark)
                for (int e = s0 + tid; e < e0; e += 256 * 4) {        // four (column, value) pairs per thread in flight, as in the gather above
                    int u[4], hu[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] = e + 256 * j < e0 ? col[e + 256 * j] : v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hu[j] = u[j] != v ? h[u[j]] : 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (hu[j] > now && hu[j] <= c) { flag_out[u[j]] = 1; ++marks; }
                }
        }
    }
    if (marks) atomicAdd(&ctl->marks[sweep & 15], marks);
}

// h = min(degree without the self loop, cap); all flags of both sets cleared
__global__ __launch_bounds__(256) void kcore_hindex_init_kernel(int n, int cap, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                                                int32_t *__restrict__ h, uint8_t *__restrict__ flags)
{
    const int lig = threadIdx.x & 7;
    const int64_t v = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3;
    if (v >= n) return;
    int cnt = 0;
    for (int e = row_ptr[v] + lig, end = row_ptr[v + 1]; e < end; e += 8) cnt += (col[e] != (int)v);
    cnt += __shfl_xor(cnt, 1, 8);
    cnt += __shfl_xor(cnt, 2, 8);
    cnt += __shfl_xor(cnt, 4, 8);
    if (lig == 0) { h[v] = min(cnt, cap); flags[v] = 0; flags[(size_t)n + v] = 0; }
}

__global__ __launch_bounds__(256) void kcore_hindex_finish_kernel(int n, const int32_t *__restrict__ h, int32_t *__restrict__ core, KhCtl *ctl)
{
    __shared__ int s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int c = 0;
    if (v < n) { c = h[v]; core[v] = c; }
#pragma unroll
    for (int o = 32; o; o >>= 1) c = max(c, __shfl_xor(c, o));
    if ((threadIdx.x & 63) == 0 && c > 0) atomicMax(&s_max, c);
    __syncthreads();
    // one device atomic per block, and only from blocks that can raise the maximum (15 600 waves on one address took 180 us)
    if (threadIdx.x == 0 && s_max > __hip_atomic_load(&ctl->max_core, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ctl->max_core, s_max);
}

// ------------------------------------------------------------------ edge levels + histogram
constexpr int EL_HIST = 2048;

__global__ __launch_bounds__(256) void edge_level_kernel(int64_t n, const int32_t *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ col, const float *__restrict__ val,
                                                         const int32_t *__restrict__ core, int32_t *__restrict__ level,
                                                         unsigned long long *count, double *wsum, int hist_len)
{
    __shared__ unsigned long long h_cnt[EL_HIST];
    __shared__ double h_sum[EL_HIST];
    const bool want_hist = (count || wsum) && hist_len > 0;
    const bool lds_hist = want_hist && hist_len <= EL_HIST;
    if (lds_hist)
        for (int i = threadIdx.x; i < hist_len; i += 256) { h_cnt[i] = 0; h_sum[i] = 0.0; }
    __syncthreads();
    const int lig = threadIdx.x & 15;
    const int64_t rows_per_block = 16;
    // grid-stride over rows so that the LDS histogram is flushed once per block
    for (int64_t row = (int64_t)blockIdx.x * rows_per_block + (threadIdx.x >> 4); row < n;
         row += (int64_t)gridDim.x * rows_per_block) {
        const int cr = core[row];
        for (int e = row_ptr[row] + lig, end = row_ptr[row + 1]; e < end; e += 16) {
            const int lv = min(cr, core[col[e]]);
            level[e] = lv;
            if (want_hist) {
                const int b = min(lv, hist_len - 1);
                if (lds_hist) {
                    atomicAdd(&h_cnt[b], 1ull);
                    if (wsum) atomicAdd(&h_sum[b], (double)val[e]);
                } else {
                    if (count) atomicAdd(&count[b], 1ull);
                    if (wsum) atomicAdd(&wsum[b], (double)val[e]);
                }
            }
        }
    }
    __syncthreads();
    if (lds_hist)
        for (int i = threadIdx.x; i < hist_len; i += 256) {
            if (count && h_cnt[i]) atomicAdd(&count[i], h_cnt[i]);
            if (wsum && h_cnt[i]) atomicAdd(&wsum[i], h_sum[i]);
        }
}

// ------------------------------------------------------------------ stable per-row partition by slot
__global__ __launch_bounds__(256) void slot_reorder_kernel(int64_t n, int K, const int32_t *__restrict__ row_ptr,
                                                           const int32_t *__restrict__ col, const float *__restrict__ val,
                                                           const int32_t *__restrict__ level,
                                                           const uint8_t *__restrict__ table, int table_len,
                                                           int32_t *__restrict__ col_out, float *__restrict__ val_out,
                                                           uint8_t *__restrict__ slot_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int start = row_ptr[row], end = row_ptr[row + 1];
    int out = start;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (end - start <= 64) {
        // one chunk: keep the row in registers, sweep the slots
        const int e = start + lane;
        const bool valid = e < end;
        int c = 0, s = -1;
        float w = 0.f;
        if (valid) { c = col[e]; w = val[e]; s = table[min(level[e], table_len - 1)]; }
        unsigned long long todo = __ballot(valid);
        while (todo) {
            // smallest slot among the entries not placed yet (slots are emitted in increasing order)
            int cand = ((todo >> lane) & 1ull) ? s : 0x7fffffff;
            for (int o = 32; o; o >>= 1) cand = min(cand, __shfl_xor(cand, o));
            const bool hit = valid && s == cand;
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = out + __popcll(m & lt);
                col_out[pos] = c; val_out[pos] = w; slot_out[pos] = (uint8_t)s;
            }
            out += __popcll(m);
            todo &= ~m;
        }
        return;
    }
    for (int s = 0; s < K; ++s) {
        for (int base = start; base < end; base += 64) {
            const int e = base + lane;
            const bool hit = e < end && (int)table[min(level[e], table_len - 1)] == s;
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int pos = out + __popcll(m & lt);
                col_out[pos] = col[e]; val_out[pos] = val[e]; slot_out[pos] = (uint8_t)s;
            }
            out += __popcll(m);
        }
    }
}

// ================================================================================================
// GRU over short sequences with many independent rows (core axis K <= ~22, time axis T <= ~100; rows = nodes).
// Reference: nn.GRU(batch_first=True) + .sum(dim=1) + LayerNorm at layers.py:59-62, nn.GRU + LayerNorm at
// models.py:249-250.  The input projection GI = x·W_ihᵀ + b is a plain library GEMM done by the caller; this
// kernel owns the RECURRENT part, which MIOpen runs as thousands of small tensor ops:
//     gh = h_{t-1}·W_hhᵀ ; r = σ(GI_r + gh_r) ; z = σ(GI_z + gh_z) ; n = tanh(GI_n + r·(gh_n + b_hn))
//     h_t = n + z·(h_{t-1} − n)
// hidden = 128 fixed.  Exact fp32: v_mfma_f32_16x16x4_f32 (an fmaf chain, one rounding per product).
// Block = 8 waves; wave w owns hidden units [16w,16w+16) of all three gates and keeps its 128x48 slice of W_hhᵀ
// in 96 VGPRs as ready-made MFMA B operands for the whole (persistent) kernel.  h_{t-1} of the block's 32 rows
// lives in LDS (double buffered); each lane reads 32 consecutive k of its row with ds_read_b128 — the reduction
// index is mapped k = 32·(lane>>4) + step so that those are exactly its A operands.  Step 0 (h = 0) issues no MFMA.
// ================================================================================================
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int GRU_H = 128;
constexpr int GRU_BM = 64;           // rows per block iteration = 4 MFMA row tiles
constexpr int GRU_RT = GRU_BM / 16;
constexpr int GRU_PITCH = 132;       // floats; 528 B keeps rows 16-B aligned

struct GruArgs {
    int64_t rows;
    int32_t steps;
    const float *gi;        // [rows, steps, 3*128]  gate order r,z,n; includes b_ih (+ b_hh for r,z)
    const float *whh;       // [3*128, 128]
    const float *bhn;       // [128] or null
    const float *gamma;     // LayerNorm weight / bias [128] or null (no LayerNorm)
    const float *beta;
    float eps;
    int32_t reduce_sum;     // 1: out[rows,128] = LN(sum_t h_t)   0: out[rows,steps,128] = LN(h_t)
    float *out;
    float *gates;           // optional [rows, steps, 4, 128]: r, z, n, q = W_hn h + b_hn, saved for the backward kernel
    int32_t gi_blocked;     // gi is in the blocked tile layout (gi_blocked_offset); fp16x2 kernel only
    int64_t ldo;            // reduce_sum: floats between output rows (128 = dense; larger: rows of a [rows, T, 128] tensor)
    // row plan of the aggregation that produced gi through the GEMM (gru_seq_h2_kernel<reduce>, plain gi layout only; all null: gi is
    // [rows, steps, 384]).  Sequence p is written to out row order[p]; gi holds only the steps that bring a new x, compactly: tile T (64
    // sequences) starts at gi row tbase[T], sequence p owns popcount(tmask[T]) consecutive rows, and step t of it reads the row of the
    // last set bit <= t of tmask[T] (a repeated x row has the same projection: the reference multiplies it again, layers.py:59).
    const int32_t *order;
    const uint32_t *tmask;
    const int32_t *tbase;
};

// v_exp_f32 / v_rcp_f32 (1 ulp each): far inside the fp32 tolerance of the layer, a fraction of an IEEE divide's cost
// raw v_exp_f32 (2^x): __expf wraps it in range clamps (v_max ...) that these forms do not need — 2^(+big) = inf -> rcp = 0,
// 2^(-big) = 0 -> rcp(1) = 1 are exactly the saturated values
__device__ __forceinline__ float gru_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504f)); }
#ifdef CTGCN_TANH_V1      // rounds 1-5: 1 - 2 / (1 + e^{2x}) — five instructions, but an ABSOLUTE error of 2-3 ulps of 1.0 over the whole range (A/B builds)
__device__ __forceinline__ float gru_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008f)); }
#else
// tanh(x) = sign(x) (1 - e) / (1 + e), e = e^{-2|x|} in (0, 1]: the numerator is exact for e >= 0.5 and carries e's error otherwise, the
// quotient's rounding is RELATIVE to the result — about 0.5 ulp of 1.0 near zero and 1.3 at |tanh| = 0.5, against 2 and 3 for the form above
// (round 6: that form's error, added to h in every recurrent step, made the RMS error of every config 1.35 x the fp32 CPU path's; with this
// form it is 0.98 - 1.08 x).  Two more instructions per value (|x| is a source modifier, the sign one v_bfi).
__device__ __forceinline__ float gru_tanh(float x)
{
    const float e = __builtin_amdgcn_exp2f(__builtin_fabsf(x) * -2.88539008f);
    return __builtin_copysignf((1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e), x);
}
#endif

// sum over the 64 lanes of a wave, result in every lane: four DPP exchanges inside the 16-lane rows (quad swaps, half-row and row mirrors —
// plain VALU, no LDS crossbar), two row broadcasts and a v_readlane, instead of six ds_bpermute round trips.  The LayerNorm of a row is two such
// sums one after the other; the per-step form of the GRU layer kernel normalises two rows per wave and unit.
__device__ __forceinline__ float wave_sum64(float v)
{
    auto dpp = [](float x, auto ctrl) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, true)); };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror: every lane holds the sum of its row of 16
    // rows 1, 3 take row 0 / 2's sum (row_bcast:15), rows 2, 3 then take the sum of rows 0-1 (row_bcast:31): lane 63 holds the total, which
    // v_readlane hands to every lane through a scalar register — no LDS crossbar at all (two ds_bpermute exchanges here cost the per-step
    // GRU layer kernel 6 %: 12.2 -> 11.4 ms per 1 M x 16 call; the six of the first version 15 %)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float2 gru_layernorm_vals(float2 v, int lane, const float *gamma, const float *beta, float eps)
{
    if (gamma) {
        const float s = wave_sum64(v.x + v.y);
        const float mean = s * (1.0f / GRU_H);
        const float dx = v.x - mean, dy = v.y - mean;
        const float q = wave_sum64(dx * dx + dy * dy);
        const float rstd = rsqrtf(q * (1.0f / GRU_H) + eps);
        const float2 g = *(const float2 *)(gamma + lane * 2);
        const float2 b = beta ? *(const float2 *)(beta + lane * 2) : float2{0.f, 0.f};
        v.x = dx * rstd * g.x + b.x;
        v.y = dy * rstd * g.y + b.y;
    }
    return v;
}
__device__ __forceinline__ void gru_layernorm_row(const float *__restrict__ src, float *__restrict__ dst, int lane,
                                                  const float *gamma, const float *beta, float eps)
{
    *(float2 *)(dst + lane * 2) = gru_layernorm_vals(*(const float2 *)(src + lane * 2), lane, gamma, beta, eps);
}

// ------------------------------------------------------------------------------------------------
// Backward of the LayerNorm behind a GRU (layers.py:61-62 norm(output.sum(dim=1)), models.py:250 norm(output)): one wave per row of 128.
//   x = sum over `steps` rows of h (steps = 1: the row itself), xhat = (x - mean) rstd, dxhat = dy gamma,
//   dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat));  dgamma = sum_rows dy xhat;  dbeta = sum_rows dy.
// The framework's LayerNorm backward on 128-wide rows (three kernels incl. a partial gamma/beta reduction) plus the sum over steps and the
// forward recompute cost the config-5 training step 265 of 1 780 ms; this is one pass: steps x 512 B + 512 B read, 512 B written per row.
// dgamma / dbeta leave as per-block partial sums [gridDim.x][256] (gamma | beta), added by the caller: deterministic.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(int64_t rows, int32_t steps, const float *__restrict__ h, const float *__restrict__ dy,
                                                            int64_t ld_dy, const float *__restrict__ gamma, float eps, float *__restrict__ dx,
                                                            float *__restrict__ partial, const int32_t *__restrict__ dy_rows)
{
    __shared__ float red[4][2 * GRU_H];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float2 g = *(const float2 *)(gamma + lane * 2);
    float2 dg = {0.f, 0.f}, db = {0.f, 0.f};
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float *src = h + row * steps * GRU_H + lane * 2;
        float2 x = *(const float2 *)src;
        for (int t = 1; t < steps; ++t) {
            const float2 v = *(const float2 *)(src + (int64_t)t * GRU_H);
            x.x += v.x; x.y += v.y;
        }
        // dy_rows: row p of h / dx is row dy_rows[p] of dy (the GRU ran in the aggregation's row-plan order, its output left un-permuted)
        const float2 d = *(const float2 *)(dy + (dy_rows ? (int64_t)dy_rows[row] : row) * ld_dy + lane * 2);
        const float s = wave_sum64(x.x + x.y);
        const float mean = s * (1.0f / GRU_H);
        const float cx = x.x - mean, cy = x.y - mean;
        const float q = wave_sum64(cx * cx + cy * cy);
        const float rstd = rsqrtf(q * (1.0f / GRU_H) + eps);
        const float hx = cx * rstd, hy = cy * rstd;
        const float ex = d.x * g.x, ey = d.y * g.y;                 // dxhat
        const float m1 = wave_sum64(ex + ey) * (1.0f / GRU_H), m2 = wave_sum64(ex * hx + ey * hy) * (1.0f / GRU_H);
        *(float2 *)(dx + row * GRU_H + lane * 2) = float2{rstd * (ex - m1 - hx * m2), rstd * (ey - m1 - hy * m2)};
        dg.x += d.x * hx; dg.y += d.y * hy;
        db.x += d.x; db.y += d.y;
    }
    *(float2 *)(&red[wave][lane * 2]) = dg;
    *(float2 *)(&red[wave][GRU_H + lane * 2]) = db;
    __syncthreads();
    const int c = threadIdx.x;                                      // 256 threads = 256 partial columns
    partial[(int64_t)blockIdx.x * (2 * GRU_H) + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

template <bool REDUCE, bool SAVE>
__global__ __launch_bounds__(512, 2) void gru_seq_kernel(const GruArgs a)
{
    // h_{t-1} / h_t (double buffered) and the running sum over steps, all [row][hidden] with a padded pitch
    __shared__ float hbuf[2][GRU_BM][GRU_PITCH];
    __shared__ float sbuf[REDUCE ? GRU_BM : 1][GRU_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int hid = wave * 16 + col;
    const int steps = a.steps;

    // B operands: W[g][kk] = W_hh[g*128 + hid][32*grp + kk]
    float W[3][32];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const f4v *src = (const f4v *)(a.whh + (int64_t)(g * GRU_H + hid) * GRU_H + 32 * grp);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f4v v = src[q];
            W[g][4 * q + 0] = v.x; W[g][4 * q + 1] = v.y; W[g][4 * q + 2] = v.z; W[g][4 * q + 3] = v.w;
        }
    }
    const float b_hn = a.bhn ? a.bhn[hid] : 0.f;
    const int64_t ntiles = (a.rows + GRU_BM - 1) / GRU_BM;
    const int gstride = steps * 3 * GRU_H;              // GI elements per row

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * GRU_BM;
        const float *gi_tile = a.gi + row0 * gstride + hid;
        const int last = (int)min((int64_t)GRU_BM, a.rows - row0) - 1;   // rows beyond the end re-read the last valid row
        // this lane's C-layout rows of row tile rt are rt*16 + grp*4 + i; 32-bit element offset into gi_tile
        auto goff = [&](int rt, int i) { return min(rt * 16 + grp * 4 + i, last) * gstride; };

        // ---- step 0: h_{-1} = 0, so gh = 0 and no MFMA is issued
#pragma unroll
        for (int rt = 0; rt < GRU_RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float *p = gi_tile + goff(rt, i);
                const float r = gru_sigmoid(p[0]);
                const float z = gru_sigmoid(p[GRU_H]);
                const float n = gru_tanh(p[2 * GRU_H] + r * b_hn);
                const float h = n - z * n;
                hbuf[0][rt * 16 + grp * 4 + i][hid] = h;
                if (REDUCE) sbuf[rt * 16 + grp * 4 + i][hid] = h;
                if (SAVE && rt * 16 + grp * 4 + i <= last) {
                    float *gp = a.gates + ((row0 + rt * 16 + grp * 4 + i) * steps) * (4 * GRU_H) + hid;
                    gp[0] = r; gp[GRU_H] = z; gp[2 * GRU_H] = n; gp[3 * GRU_H] = b_hn;
                }
            }
        __syncthreads();
        if (!REDUCE)
            for (int r = wave; r <= last; r += 8)
                gru_layernorm_row(hbuf[0][r], a.out + ((row0 + r) * steps) * GRU_H, lane, a.gamma, a.beta, a.eps);

        // ---- steps 1..: software pipeline over the row tiles — the MFMAs of tile rt run with the gate math of tile rt-1
        for (int t = 1; t < steps; ++t) {
            const float(*hprev)[GRU_PITCH] = hbuf[(t - 1) & 1];
            float(*hcur)[GRU_PITCH] = hbuf[t & 1];
            const float *gi_t = gi_tile + t * 3 * GRU_H;
            f4v acc[2][3];
            float gi[2][3][4];
#pragma unroll
            for (int rt = 0; rt <= GRU_RT; ++rt) {
                const int cur = rt & 1, prv = cur ^ 1;
                float av[32];
                if (rt < GRU_RT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int g = 0; g < 3; ++g) gi[cur][g][i] = gi_t[goff(rt, i) + g * GRU_H];   // used one stage later
                    const f4v *src = (const f4v *)(&hprev[rt * 16 + col][32 * grp]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f4v v = src[q];
                        av[4 * q + 0] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
                    }
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[cur][g] = f4v{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    if (rt < GRU_RT) {
#pragma unroll
                        for (int g = 0; g < 3; ++g)
                            acc[cur][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], W[g][kk], acc[cur][g], 0, 0, 0);
                    }
                    if (rt > 0 && (kk & 7) == 4) {       // one element of the previous tile's gates every 8 k-steps
                        const int i = kk >> 3;
                        const int r_ = (rt - 1) * 16 + grp * 4 + i;
                        const float hold = hprev[r_][hid];
                        const float r = gru_sigmoid(gi[prv][0][i] + acc[prv][0][i]);
                        const float z = gru_sigmoid(gi[prv][1][i] + acc[prv][1][i]);
                        const float n = gru_tanh(gi[prv][2][i] + r * (acc[prv][2][i] + b_hn));
                        const float h = n + z * (hold - n);
                        hcur[r_][hid] = h;
                        if (REDUCE) sbuf[r_][hid] += h;
                        if (SAVE && r_ <= last) {
                            float *gp = a.gates + ((row0 + r_) * steps + t) * (4 * GRU_H) + hid;
                            gp[0] = r; gp[GRU_H] = z; gp[2 * GRU_H] = n; gp[3 * GRU_H] = acc[prv][2][i] + b_hn;
                        }
                    }
                }
            }
            __syncthreads();
            if (!REDUCE)
                for (int r = wave; r <= last; r += 8)
                    gru_layernorm_row(hcur[r], a.out + ((row0 + r) * steps + t) * GRU_H, lane, a.gamma, a.beta, a.eps);
        }
        if (REDUCE)
            for (int r = wave; r <= last; r += 8)
                gru_layernorm_row(sbuf[r], a.out + (row0 + r) * a.ldo, lane, a.gamma, a.beta, a.eps);
        __syncthreads();       // LDS is reused by the next tile
    }
}

// ================================================================================================
// fp32 GEMM arithmetic on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 terms
// (x = x1 + x2 + x3, 8 + 8 + 8 = 24 mantissa bits; the residuals x - x1 and x - x1 - x2 are exact in fp32) and a
// product is the six partial products x1y1 + x1y2 + x2y1 + x1y3 + x3y1 + x2y2 (everything down to 2^-24 relative),
// each exact in fp32 and accumulated in the MFMA's fp32 accumulator.  Measured max error vs fp64 on a K = 128 dot
// product: 1.8e-6 against 3.1e-6 for the fp32 fmaf chain (tools/probes/mfma_bf16x3_probe.hip) — fp32 accuracy at
// ~2x the fp32-MFMA rate (v_mfma_f32_16x16x32_bf16 is 16x faster than v_mfma_f32_16x16x4_f32, 6 of them per product).
// ================================================================================================
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bf16_split3(float x, __bf16 &a, __bf16 &b, __bf16 &c)
{
    a = (__bf16)x;
    const float r1 = x - (float)a;
    b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    c = (__bf16)r2;
}

// 8 consecutive fp32 -> three bf16x8 fragments
__device__ __forceinline__ void bf16_split3_x8(const float *src, bf8v &s0, bf8v &s1, bf8v &s2)
{
    const f4v lo = *(const f4v *)src, hi = *(const f4v *)(src + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 a, b, c;
        bf16_split3(lo[j], a, b, c); s0[j] = a; s1[j] = b; s2[j] = c;
        bf16_split3(hi[j], a, b, c); s0[4 + j] = a; s1[4 + j] = b; s2[4 + j] = c;
    }
}

// the six partial products, smallest magnitude first; index pairs (split of A, split of B)
#define CTGCN_X3_PAIRS(F) F(2, 0) F(0, 2) F(1, 1) F(1, 0) F(0, 1) F(0, 0)

// ------------------------------------------------------------------------------------------------
// GRU input projection  GI[rows, 384] = X[rows, 128] · W_ihᵀ + bias   (the `gi` operand of gru_seq_kernel).
// Same block shape as the recurrence: wave w owns output columns {g*128 + 16w + c} and keeps its slice of W_ihᵀ,
// already split, in 144 VGPRs (36 bf16x8 B fragments).  The 64-row X tile is split once by the whole block and
// staged in LDS as three bf16 planes (double buffered: the next tile's global loads fly during the MFMAs).
// HBM-bound once the matrix pipe is this fast: 512 B read + 1536 B written per row.
// ------------------------------------------------------------------------------------------------
constexpr int PJ_BM = 64;
constexpr int PJ_PITCH = GRU_H + 8;      // bf16 elements; 272-byte rows keep ds_read_b128 conflict-free

struct ProjArgs {
    int64_t rows;
    const float *x;
    int64_t ldx;
    const float *w;       // [384, 128]
    const float *bias;    // [384] or null
    float *out;           // [rows, 384], or the blocked layout below
    int32_t steps_blocked; // > 0: rows = nodes*steps sequences; out is written in the recurrence's tile layout (gi_blocked_offset)
};

// Blocked GI layout (fp16x2 path): [node tile of 64][step][gate][wave = 16-column group][node in tile][16 columns] floats.
// The recurrence kernel's wave-load for (step, gate, 16-row tile) is then ONE contiguous KB instead of sixteen 64-byte
// pieces 12 KB apart (measured: recurrence 3.32 -> 3.01 ms per 1 M x 8 call).  Same size as [nodes, steps, 384] when the
// node count is a multiple of 64; the buffer must cover whole tiles.
__device__ __forceinline__ int64_t gi_blocked_offset(int64_t node, int t, int steps, int gate, int column)
{
    return ((((node >> 6) * steps + t) * 3 + gate) * 8 + (column >> 4)) * 1024 + (node & 63) * 16 + (column & 15);
}

__global__ __launch_bounds__(512, 2) void gru_proj_x3_kernel(const ProjArgs a)
{
    __shared__ __bf16 As[2][3][PJ_BM][PJ_PITCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int hid = wave * 16 + col;

    bf8v Wf[3][4][3];      // [split][k chunk][gate]
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            bf16_split3_x8(a.w + (int64_t)(g * GRU_H + hid) * GRU_H + c * 32 + 8 * grp, Wf[0][c][g], Wf[1][c][g], Wf[2][c][g]);
    // The MFMA is issued transposed (weights as the A operand, X rows as the B operand): D[m = out column][n = X row],
    // so a lane ends up with FOUR CONSECUTIVE output columns 16w + 4*grp .. +3 of X row (lane & 15): float4 I/O.
    const int oc = wave * 16 + 4 * grp;
    f4v bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = a.bias ? *(const f4v *)(a.bias + g * GRU_H + oc) : f4v{0.f, 0.f, 0.f, 0.f};

    const int64_t ntiles = (a.rows + PJ_BM - 1) / PJ_BM;
    // staging role: 64 rows x 32 float4 = 2048 float4, four per thread; idx -> (row, c4)
    auto load_tile = [&](int64_t tile, f4v (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            int64_t r = tile * PJ_BM + (idx >> 5);
            r = r < a.rows ? r : a.rows - 1;
            v[i] = *(const f4v *)(a.x + r * a.ldx + (idx & 31) * 4);
        }
    };
    auto stage_tile = [&](int buf, const f4v (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            bf4v s0, s1, s2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 p, q, r;
                bf16_split3(v[i][j], p, q, r);
                s0[j] = p; s1[j] = q; s2[j] = r;
            }
            __bf16 *dst = &As[buf][0][idx >> 5][(idx & 31) * 4];
            *(bf4v *)dst = s0;
            *(bf4v *)(dst + PJ_BM * PJ_PITCH) = s1;
            *(bf4v *)(dst + 2 * PJ_BM * PJ_PITCH) = s2;
        }
    };

    f4v stage[4];
    int buf = 0;
    if ((int64_t)blockIdx.x < ntiles) {
        load_tile(blockIdx.x, stage);
        stage_tile(0, stage);
    }
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load_tile(next, stage);               // in flight during the MFMAs
        const int64_t row0 = tile * PJ_BM;
#pragma unroll
        for (int rt = 0; rt < PJ_BM / 16; ++rt) {
            f4v acc[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf8v af[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) af[sp] = *(const bf8v *)(&As[buf][sp][rt * 16 + col][c * 32 + 8 * grp]);
#define CTGCN_X3_MFMA(I, J)                                                                                          \
                _Pragma("unroll") for (int g = 0; g < 3; ++g)                                                        \
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[J][c][g], af[I], acc[g], 0, 0, 0);
                CTGCN_X3_PAIRS(CTGCN_X3_MFMA)
#undef CTGCN_X3_MFMA
            }
            const int64_t row = row0 + rt * 16 + col;
            if (row < a.rows) {
                float *o = a.out + row * (3 * GRU_H) + oc;
#pragma unroll
                for (int g = 0; g < 3; ++g) *(f4v *)(o + g * GRU_H) = acc[g] + bias[g];
            }
        }
        if (next < ntiles) stage_tile(buf ^ 1, stage);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// gru_seq_x3_kernel: the REDUCE form of gru_seq_kernel (LayerNorm(sum_t h_t), the CoreDiffusion case = 32 of the 33
// GRU calls of a CTGCN-C window) with the recurrent product h_{t-1}·W_hhᵀ in split-bf16 arithmetic (see above).
// W_hhᵀ slice: 36 bf16x8 fragments (144 VGPRs) per wave; h_t is split by the lane that produces it and stored as
// three bf16 planes in LDS (double buffered), from which the A fragments are single ds_read_b128; the fp32 h needed
// by the z·h_{t-1} term stays in registers, the running sum in an fp32 LDS plane.
// ------------------------------------------------------------------------------------------------
template <bool REDUCE, bool SAVE>
__global__ __launch_bounds__(512, 2) void gru_seq_x3_kernel(const GruArgs a)
{
    __shared__ __bf16 Hs[2][3][GRU_BM][PJ_PITCH];
    __shared__ float sbuf[GRU_BM][GRU_PITCH];      // REDUCE: running sum over steps; otherwise: fp32 h_t staged for the row-wise LayerNorm / store
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int hid = wave * 16 + col;          // weight row this lane holds fragments of (MFMA m index)
    const int oc = wave * 16 + 4 * grp;       // first of the 4 consecutive hidden units this lane produces
    const int steps = a.steps;

    // MFMA A operand (transposed product D[m = hidden][n = row]):  A[m = hid][k = c*32 + 8*grp + j] = W_hh[g*128 + hid][k]
    bf8v Wf[3][4][3];      // [split][k chunk][gate]
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            bf16_split3_x8(a.whh + (int64_t)(g * GRU_H + hid) * GRU_H + c * 32 + 8 * grp, Wf[0][c][g], Wf[1][c][g], Wf[2][c][g]);
    const f4v b_hn = a.bhn ? *(const f4v *)(a.bhn + oc) : f4v{0.f, 0.f, 0.f, 0.f};
    const int64_t ntiles = (a.rows + GRU_BM - 1) / GRU_BM;
    const int gstride = steps * 3 * GRU_H;

    for (int64_t tile_ = blockIdx.x; tile_ < ntiles; tile_ += gridDim.x) {
        const int64_t tile = ntiles - 1 - tile_;      // newest GI first: the projection kernel wrote the high tiles last
        const int64_t row0 = tile * GRU_BM;
        const float *gi_tile = a.gi + row0 * gstride + oc;
        const int last = (int)min((int64_t)GRU_BM, a.rows - row0) - 1;
        // a lane's results: row rt*16 + col, hidden oc..oc+3 (float4 / bf16x4 I/O everywhere)
        auto goff = [&](int rt) { return min(rt * 16 + col, last) * gstride; };
        f4v hreg[GRU_RT];

        auto publish = [&](int buf, int r_, const f4v h) {     // split h and store the three bf16 planes
            bf4v p, q, r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 x, y, z;
                bf16_split3(h[j], x, y, z);
                p[j] = x; q[j] = y; r[j] = z;
            }
            *(bf4v *)(&Hs[buf][0][r_][oc]) = p;
            *(bf4v *)(&Hs[buf][1][r_][oc]) = q;
            *(bf4v *)(&Hs[buf][2][r_][oc]) = r;
        };
        // gate math for the lane's 4 hidden units of one row; REDUCE == false also emits the raw h (and, SAVE, the gates)
        auto gates = [&](const f4v gr, const f4v gz, const f4v gn, const f4v ar, const f4v az, const f4v an, const f4v hold,
                         int t, int r_) {
            f4v h, rv, zv, nv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rv[j] = gru_sigmoid(gr[j] + ar[j]);
                zv[j] = gru_sigmoid(gz[j] + az[j]);
                nv[j] = gru_tanh(gn[j] + rv[j] * (an[j] + b_hn[j]));
                h[j] = nv[j] + zv[j] * (hold[j] - nv[j]);
            }
            if (!REDUCE) *(f4v *)(&sbuf[r_][oc]) = h;
            if (SAVE && r_ <= last) {
                float *gp = a.gates + ((row0 + r_) * steps + t) * (4 * GRU_H) + oc;
                *(f4v *)gp = rv; *(f4v *)(gp + GRU_H) = zv; *(f4v *)(gp + 2 * GRU_H) = nv; *(f4v *)(gp + 3 * GRU_H) = an + b_hn;
            }
            return h;
        };
        const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};

        // ---- step 0: h_{-1} = 0, no MFMA
#pragma unroll
        for (int rt = 0; rt < GRU_RT; ++rt) {
            const float *p = gi_tile + goff(rt);
            const f4v h = gates(*(const f4v *)p, *(const f4v *)(p + GRU_H), *(const f4v *)(p + 2 * GRU_H), zero4, zero4, zero4, zero4, 0, rt * 16 + col);
            hreg[rt] = h;
            publish(0, rt * 16 + col, h);
            if (REDUCE) *(f4v *)(&sbuf[rt * 16 + col][oc]) = h;
        }
        __syncthreads();
        auto emit_step = [&](int t) {      // per-step output: LayerNorm (or plain copy) of the staged fp32 rows, 512 B per row
            for (int r = wave; r <= last; r += 8)
                gru_layernorm_row(sbuf[r], a.out + ((row0 + r) * steps + t) * GRU_H, lane, a.gamma, a.beta, a.eps);
            __syncthreads();               // the next step's gate math overwrites sbuf
        };
        if (!REDUCE) emit_step(0);

        // Software pipeline over the row tiles of a step: the gate math of tile rt-1 (a ~100-instruction VALU block) is
        // issued in the shadow of tile rt's MFMAs instead of stalling the matrix pipe behind them.
        for (int t = 1; t < steps; ++t) {
            const int pb = (t - 1) & 1, cb = t & 1;
            const float *gi_t = gi_tile + t * 3 * GRU_H;
            f4v acc[2][3], gq[2][3];
#pragma unroll
            for (int rt = 0; rt <= GRU_RT; ++rt) {
                const int cur = rt & 1, prv = cur ^ 1;
                if (rt < GRU_RT) {
                    const float *p = gi_t + goff(rt);
                    gq[cur][0] = *(const f4v *)p; gq[cur][1] = *(const f4v *)(p + GRU_H); gq[cur][2] = *(const f4v *)(p + 2 * GRU_H);
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[cur][g] = zero4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        bf8v af[3];      // MFMA B operand: B[k][n = row] = h_{t-1}[row rt*16 + col][k = c*32 + 8*grp + j]
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) af[sp] = *(const bf8v *)(&Hs[pb][sp][rt * 16 + col][c * 32 + 8 * grp]);
#define CTGCN_X3_MFMA(I, J)                                                                                              \
                        _Pragma("unroll") for (int g = 0; g < 3; ++g)                                                    \
                            acc[cur][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[J][c][g], af[I], acc[cur][g], 0, 0, 0);
                        CTGCN_X3_PAIRS(CTGCN_X3_MFMA)
#undef CTGCN_X3_MFMA
                    }
                }
                if (rt > 0) {
                    const int rp = rt - 1;
                    const f4v h = gates(gq[prv][0], gq[prv][1], gq[prv][2], acc[prv][0], acc[prv][1], acc[prv][2], hreg[rp], t, rp * 16 + col);
                    hreg[rp] = h;
                    publish(cb, rp * 16 + col, h);
                    if (REDUCE) {
                        f4v *sp_ = (f4v *)(&sbuf[rp * 16 + col][oc]);
                        *sp_ = *sp_ + h;
                    }
                }
            }
            __syncthreads();
            if (!REDUCE) emit_step(t);
        }
        if (REDUCE)
            for (int r = wave; r <= last; r += 8)
                gru_layernorm_row(sbuf[r], a.out + (row0 + r) * a.ldo, lane, a.gamma, a.beta, a.eps);
        __syncthreads();       // LDS is reused by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// fp16x2 split arithmetic (the default of the forward path).  Measured (tools/probes/mfma_f16x2_probe.hip): an fp32
// operand x is written x = s·(x1 + x2·2^-11) with s a power of two chosen per ROW so that max|x/s| is in [2^14, 2^15),
// x1 = fp16(x/s), x2 = fp16((x/s − x1)·2^11)  (22 mantissa bits, no subnormal loss: the residual is rescaled).  Then
//     x·y  ≈  sx·sy·[ x1·y1 + 2^-11·(x1·y2 + x2·y1) ]           (dropped: x2·y2 ≈ 2^-22 x·y)
// = THREE v_mfma_f32_16x16x32_f16 into two fp32 accumulators.  Max error / Σ|a·b| over K = 128 dot products:
// 1.1e-7 (projection-like data), against 2.5e-7 for the 3-way bf16 split with six products and 2.0e-7 for an fp32 fmaf
// chain — more accurate than both at HALF the matrix-core work of bf16x3, two operand planes instead of three in LDS
// and 96 instead of 144 weight VGPRs.  On gfx950 VALU work does not overlap MFMA work on a SIMD (tools/probes/
// mfma_valu_overlap_probe.hip: MFMA + k FMAs costs the SUM), so halving the MFMAs is a direct win.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// power-of-two scale s with m/s in [2^14, 2^15) for m = max|row| (biased exponent arithmetic; m = 0 or tiny -> a harmless huge 1/s)
__device__ __forceinline__ void h2_scale(float m, float &s, float &inv_s)
{
    int e = (int)(__float_as_uint(m) >> 23);          // m >= 0
    e = e < 15 ? 15 : (e > 253 ? 253 : e);
    s = __uint_as_float((uint32_t)(e - 14) << 23);
    inv_s = __uint_as_float((uint32_t)(268 - e) << 23);
}
// RS = residual scale.  Every forward kernel uses RS = 1: the three products of a term pair go into ONE fp32 accumulator (the residual of
// an element below 2^-18 of its row's maximum loses bits — an absolute error of 2^-40 of that maximum; measured 1.2e-7 vs 3.0e-7
// for an fp32 chain).  RS = 2048 (residual exact down to fp16's subnormals, needs a second accumulator and one more FMA per output)
// was the first form of the x·W_ih products; it cost the matrix-core-bound layer kernel 12 VALU instructions per unit.
template <int RS>
__device__ __forceinline__ void h2_split(float xs, _Float16 &a, _Float16 &b)     // xs = x / s
{
    a = (_Float16)xs;
    b = (_Float16)((xs - (float)a) * (float)RS);
}
// maximum over aligned groups of 32 (or 64) lanes, result in every lane: four DPP exchanges inside the 16-lane rows (quad swaps,
// half-row and row mirrors: plain VALU, no LDS crossbar) and one (two) ds_bpermute across rows.  A slot closes for every (row, core)
// pair — five dependent ds_bpermute round trips there cost the d = 128 aggregation 19 % (1.49 -> 1.77 ms per 1 M-row launch).
template <int LPR>
__device__ __forceinline__ float group_max(float m)       // m >= 0: its bit pattern orders like an unsigned integer (no NaN canonicalisation per step)
{
    uint32_t u = __float_as_uint(m);
    auto dpp = [](uint32_t v, auto ctrl) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, decltype(ctrl)::value, 0xf, 0xf, true); };
    u = max(u, dpp(u, std::integral_constant<int, 0xB1>{}));      // quad_perm [1,0,3,2]
    u = max(u, dpp(u, std::integral_constant<int, 0x4E>{}));      // quad_perm [2,3,0,1]
    u = max(u, dpp(u, std::integral_constant<int, 0x141>{}));     // row_half_mirror: lane i <-> 7 - i of its 8
    u = max(u, dpp(u, std::integral_constant<int, 0x140>{}));     // row_mirror: lane i <-> 15 - i of its 16
    u = max(u, (uint32_t)__shfl_xor((int)u, 16, 64));
    if (LPR == 64) u = max(u, (uint32_t)__shfl_xor((int)u, 32, 64));
    return __uint_as_float(u);
}

// ------------------------------------------------------------------------------------------------
// agg_fwd_split_kernel: the CoreDiffusion aggregation whose consumer is the split GEMM (ctgcn_gemm.hip) — the GRU input
// projection of a layer with d_in != 128.  Same recurrence as agg_fwd_kernel, but ONE wave holds the whole feature row
// (CH float4 chunks per lane, d <= 256 CH), so when a slot closes the row's maximum is a wave reduction away and the row
// leaves as the GEMM's operand: per-row power-of-two scale + two fp16 planes [n K, kp] (exactly what split_rows_h2_kernel
// would make of the fp32 row, bit for bit).  LPR lanes per row: 64 (d > 128: one wave per row) or 32 (d <= 128: two rows per wave,
// the mapping of agg_fwd_kernel at d = 128).  The fp32 H [n, K, d] is never written and never read back: per (row, slot)
// 4 d bytes written instead of 4 d written + 4 d read + 4 d written, and one pass over the entries instead of d / 256.
// ------------------------------------------------------------------------------------------------
// WIDE (round 5): a row plan over 33-64 slots: two mask words per tile.  Separate instantiations: the K <= 32 kernels keep their code.
template <int LPR, int CH, int U, bool WIDE = false>
__device__ __forceinline__ void agg_fwd_split_body(const AggArgs &a, _Float16 *__restrict__ p1, _Float16 *__restrict__ p2,
                                                   float *__restrict__ scale, int32_t kp, float residual_scale, const int64_t bid)
{
    // bid: index of this block among the blocks of `a` (blockIdx.x, or — grouped launch of a window — blockIdx.x % blocks per snapshot)
    const int lig = threadIdx.x & (LPR - 1);
    const int64_t pos = bid * (256 / LPR) + (threadIdx.x / LPR);
    if (pos >= a.n) return;
    const int64_t row = a.order ? (int64_t)a.order[pos] : pos;
    // the block's 256 / LPR positions lie in one 16-position tile: a scalar load
    const int64_t tile = (bid * (256 / LPR)) >> a.tile_shift;
    // step mask of the tile: one word, or (WIDE) two: tmask[2 tile] = slots 0-31, tmask[2 tile + 1] = slots 32-63
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type mask_t;
    constexpr int MB = WIDE ? 63 : 31;
    mask_t need;
    int nneed;
    if constexpr (WIDE) {
        need = a.tmask ? ((mask_t)a.tmask[2 * tile] | ((mask_t)a.tmask[2 * tile + 1] << (WIDE ? 32 : 0))) : ~(mask_t)0;
        nneed = __popcll(need);
    } else {
        need = a.tmask ? a.tmask[tile] : 0xffffffffu;
        nneed = __popc(need);
    }
    const int64_t obase = a.tbase ? (int64_t)a.tbase[tile] + (pos & ((1 << a.tile_shift) - 1)) * nneed : pos * a.K;
    const int start = a.row_ptr[row], end = a.row_ptr[row + 1];
    if (end - start > a.long_thresh) return;          // hub row: agg_fwd_hub_kernel into the compact scratch, split afterwards
    const bool self = (a.flags & CTGCN_F_SELF_LOOP) != 0;
    const bool relu = (a.flags & CTGCN_F_RELU) != 0;
    const bool nested = (a.flags & CTGCN_F_NESTED) != 0;
    const uint8_t *__restrict__ slot = a.slot;
    const float *__restrict__ X = a.src;
    bool live[CH];
    int64_t foff[CH];
    f4 R[CH], P[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = c * LPR + lig;
        live[c] = ch < a.chunks;
        foff[c] = live[c] ? (int64_t)ch * 4 : 0;      // dead lanes read chunk 0 (valid memory): every load unconditional
        R[c] = P[c] = vzero<4>();
        if (self && live[c]) R[c] = *(const f4 *)(X + row * a.ldsrc + foff[c]);
    }
    int cur = 0;

    auto close_slot = [&]() {
        f4 v[CH];
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            R[c] += P[c];
            if (!nested) P[c] = vzero<4>();
        }
        if (!((need >> (cur & MB)) & 1)) { ++cur; return; }      // a repeat of the previous slot's row that the consumer never reads
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            v[c] = relu ? vmax0(R[c]) : R[c];
            if (!live[c]) v[c] = vzero<4>();
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[c].x), fabsf(v[c].y)), fmaxf(fabsf(v[c].z), fabsf(v[c].w))));
        }
        m = group_max<LPR>(m);
        float s, inv;
        h2_scale(m, s, inv);
        int64_t orow;                                 // compact layout: the set bits of `need` below slot cur
        if constexpr (WIDE) orow = obase + (a.tbase ? __popcll(need & (((mask_t)1 << (cur & 63)) - (mask_t)1)) : cur);
        else orow = obase + (a.tbase ? __popc(need & ((1u << cur) - 1u)) : cur);
        if (lig == 0) scale[orow] = s;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int k = (c * LPR + lig) * 4;
            if (k < kp) {                             // columns [d, kp) are written as zeros
                const float xs[4] = {v[c].x * inv, v[c].y * inv, v[c].z * inv, v[c].w * inv};
                h4v h1, h2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h1[j] = (_Float16)xs[j];
                    h2[j] = (_Float16)((xs[j] - (float)h1[j]) * residual_scale);      // 1 for both consumers (kept as a parameter: the split is defined by (scale, residual scale))
                }
                __builtin_nontemporal_store(h1, (h4v *)(p1 + orow * kp + k));
                __builtin_nontemporal_store(h2, (h4v *)(p2 + orow * kp + k));
            }
        }
        ++cur;
    };

    for (int base = start; base < end; base += LPR) {
        const int my = base + lig;
        int c = 0, s = 0;
        float w = 0.f;
        if (my < end) {
            c = a.col[my];
            w = a.val[my];
            s = slot ? (int)slot[my] : 0;
        }
        const int cnt = min(LPR, end - base);
        int j = 0;
        for (; j + U <= cnt; j += U) {
            f4 xv[U][CH];
            float wj[U];
            int sj[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cj = __shfl(c, j + u, LPR);
                wj[u] = __shfl(w, j + u, LPR);
                sj[u] = __shfl(s, j + u, LPR);
#pragma unroll
                for (int q = 0; q < CH; ++q) xv[u][q] = *(const f4 *)(X + (int64_t)cj * a.ldsrc + foff[q]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                while (cur < sj[u]) close_slot();
#pragma unroll
                for (int q = 0; q < CH; ++q) P[q] = vfma(wj[u], xv[u][q], P[q]);
            }
        }
        for (; j < cnt; ++j) {
            const int cj = __shfl(c, j, LPR);
            const float w1 = __shfl(w, j, LPR);
            const int s1 = __shfl(s, j, LPR);
            f4 x1[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) x1[q] = *(const f4 *)(X + (int64_t)cj * a.ldsrc + foff[q]);
            while (cur < s1) close_slot();
#pragma unroll
            for (int q = 0; q < CH; ++q) P[q] = vfma(w1, x1[q], P[q]);
        }
    }
    while (cur < a.K) close_slot();
}

template <int LPR, int CH, int U, bool WIDE = false>
__global__ __launch_bounds__(256) void agg_fwd_split_kernel(const AggArgs a, _Float16 *__restrict__ p1, _Float16 *__restrict__ p2,
                                                            float *__restrict__ scale, int32_t kp, float residual_scale)
{
    agg_fwd_split_body<LPR, CH, U, WIDE>(a, p1, p2, scale, kp, residual_scale, blockIdx.x);
}
// d <= 128: held to the 72 registers of seven waves per SIMD, like agg_fwd_kernel<4,32,4> (left alone the epilogue takes 74: six)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 8))) void agg_fwd_split32_kernel(
    const AggArgs a, _Float16 *__restrict__ p1, _Float16 *__restrict__ p2, float *__restrict__ scale, int32_t kp, float residual_scale)
{
    agg_fwd_split_body<32, 1, 4>(a, p1, p2, scale, kp, residual_scale, blockIdx.x);
}
// small graphs (<= 200 000 rows): eight gathers in flight per lane instead of four.  A launch is then a few waves of blocks deep and its
// time is the rows' dependent load chains, not the bandwidth: 0.242 -> 0.210 ms at 87 036 rows (1 M rows: 2.18 vs 2.21 ms, occupancy 7 -> 5)
__global__ __launch_bounds__(256) void agg_fwd_split32_u8_kernel(
    const AggArgs a, _Float16 *__restrict__ p1, _Float16 *__restrict__ p2, float *__restrict__ scale, int32_t kp, float residual_scale)
{
    agg_fwd_split_body<32, 1, 8>(a, p1, p2, scale, kp, residual_scale, blockIdx.x);
}
// The d = 128 aggregation of EVERY snapshot of a window in one launch (small graphs: 25-200 us per snapshot, bound by ramp, tail and the
// rows' dependent load chains rather than by bandwidth): all snapshots share the node set, so snapshot = blockIdx.x / blocks_per_group.
struct AggSplitGroup {
    AggArgs a;
    _Float16 *p1, *p2;
    float *scale;
};
__global__ __launch_bounds__(256) void agg_fwd_split32_group_kernel(const AggSplitGroup *__restrict__ table, int32_t blocks_per_group, int32_t kp,
                                                                    float residual_scale)
{
    const int g = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (unsigned)blocks_per_group));
    const AggSplitGroup G = table[g];
    agg_fwd_split_body<32, 1, 8>(G.a, G.p1, G.p2, G.scale, kp, residual_scale, (int64_t)(blockIdx.x % (unsigned)blocks_per_group));
}
// the same for rows wider than 128 (a whole wave per row; the 500-wide first layer of the shipped configs, GEMM consumer: compact operand rows)
template <int CH>
__global__ __launch_bounds__(256) void agg_fwd_split_group_kernel(const AggSplitGroup *__restrict__ table, int32_t blocks_per_group, int32_t kp,
                                                                  float residual_scale)
{
    const int g = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (unsigned)blocks_per_group));
    const AggSplitGroup G = table[g];
    agg_fwd_split_body<64, CH, 8>(G.a, G.p1, G.p2, G.scale, kp, residual_scale, (int64_t)(blockIdx.x % (unsigned)blocks_per_group));
}

// 8 consecutive fp32 weights, already multiplied by 1/s, -> two fp16x8 fragments
template <int RS>
__device__ __forceinline__ void h2_split_x8(const float *src, float inv_s, h8v &s0, h8v &s1)
{
    const f4v lo = *(const f4v *)src, hi = *(const f4v *)(src + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        _Float16 a, b;
        h2_split<RS>(lo[j] * inv_s, a, b); s0[j] = a; s1[j] = b;
        h2_split<RS>(hi[j] * inv_s, a, b); s0[4 + j] = a; s1[4 + j] = b;
    }
}
// Weight slice of a wave for the transposed product (weights = MFMA A operand): lane (col, grp) holds, for gate g and k chunk c,
// W[g*128 + 16w + col][c*32 + 8*grp .. +7].  Rows are scaled individually: the row max is reduced over the four lanes that
// share a row (grp 0..3), the scale goes to wscale[g][16w + col] (LDS, read back per OUTPUT column by the caller).
template <int RS>
__device__ __forceinline__ void h2_load_weights(const float *w, int wave, int col, int grp, h8v (&Wf)[2][4][3], float (*wscale)[GRU_H])
{
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float *row = w + (int64_t)(g * GRU_H + wave * 16 + col) * GRU_H;
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4v lo = *(const f4v *)(row + c * 32 + 8 * grp), hi = *(const f4v *)(row + c * 32 + 8 * grp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(fabsf(lo[j]), fabsf(hi[j])));
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sc, inv;
        h2_scale(m, sc, inv);
        if (grp == 0) wscale[g][wave * 16 + col] = sc;
#pragma unroll
        for (int c = 0; c < 4; ++c) h2_split_x8<RS>(row + c * 32 + 8 * grp, inv, Wf[0][c][g], Wf[1][c][g]);
    }
}
// the three products of one k chunk for the three gates in ONE fp32 accumulator, small terms first: W1·x2, W2·x1 (both 2^-11 of the
// third), W1·x1; the residual planes are unscaled (RS = 1, see h2_split)
#define CTGCN_H2_MFMA1(WF, C, X1, X2, A)                                                                                 \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) A[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[0][C][g], X2, A[g], 0, 0, 0); \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) A[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[1][C][g], X1, A[g], 0, 0, 0); \
    _Pragma("unroll") for (int g = 0; g < 3; ++g) A[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WF[0][C][g], X1, A[g], 0, 0, 0);

// GRU input projection, fp16x2 arithmetic.  Same structure as gru_proj_x3_kernel; in addition every X row gets its own
// power-of-two scale (row max reduced over the 32 lanes that stage the row), kept in LDS next to the planes.
__global__ __launch_bounds__(512, 2) void gru_proj_h2_kernel(const ProjArgs a)
{
    __shared__ _Float16 As[2][2][PJ_BM][PJ_PITCH];
    __shared__ float rscale[2][PJ_BM];
    __shared__ float wscale[3][GRU_H];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 15, grp = lane >> 4;

    h8v Wf[2][4][3];      // [split][k chunk][gate]
    h2_load_weights<1>(a.w, wave, col, grp, Wf, wscale);
    __syncthreads();
    // transposed product: D[m = out column][n = X row]; a lane ends up with output columns 16w + 4*grp .. +3 of X row (lane & 15)
    const int oc = wave * 16 + 4 * grp;
    f4v bias[3], wsc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        bias[g] = a.bias ? *(const f4v *)(a.bias + g * GRU_H + oc) : f4v{0.f, 0.f, 0.f, 0.f};
        wsc[g] = *(const f4v *)(&wscale[g][oc]);
    }
    // Tiles: 64 consecutive rows, or — blocked output — 64 consecutive NODES at one step (rows S apart), so that a wave's
    // 16 rows x 64 bytes of output are one contiguous KB of the blocked layout.
    const int S = a.steps_blocked;
    const int64_t nodes = S > 0 ? a.rows / S : 0;
    const int64_t ntiles = S > 0 ? ((nodes + PJ_BM - 1) / PJ_BM) * S : (a.rows + PJ_BM - 1) / PJ_BM;
    auto tile_row = [&](int64_t tile, int r_) -> int64_t {      // flat row of tile row r_, clamped to the last valid one
        if (S > 0) {
            const int64_t nt = tile / S;
            const int64_t node = min(nt * PJ_BM + r_, nodes - 1);
            return node * S + (tile - nt * S);
        }
        return min(tile * PJ_BM + r_, a.rows - 1);
    };
    // staging role: 64 rows x 32 float4; idx -> (row = idx >> 5, c4 = idx & 31): the 32 lanes of a half wave hold one row
    auto load_tile = [&](int64_t tile, f4v (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            v[i] = *(const f4v *)(a.x + tile_row(tile, idx >> 5) * a.ldx + (idx & 31) * 4);
        }
    };
    auto stage_tile = [&](int buf, const f4v (&v)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            float m = fmaxf(fmaxf(fabsf(v[i][0]), fabsf(v[i][1])), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
            float sc, inv;
            h2_scale(m, sc, inv);
            if ((idx & 31) == 0) rscale[buf][idx >> 5] = sc;
            h4v s0, s1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 p, q;
                h2_split<1>(v[i][j] * inv, p, q);
                s0[j] = p; s1[j] = q;
            }
            _Float16 *dst = &As[buf][0][idx >> 5][(idx & 31) * 4];
            *(h4v *)dst = s0;
            *(h4v *)(dst + PJ_BM * PJ_PITCH) = s1;
        }
    };

    f4v stage[4];
    int buf = 0;
    if ((int64_t)blockIdx.x < ntiles) {
        load_tile(blockIdx.x, stage);
        stage_tile(0, stage);
    }
    __syncthreads();
    const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load_tile(next, stage);               // in flight during the MFMAs
        const int64_t nt = S > 0 ? tile / S : 0;
        const int tstep = S > 0 ? (int)(tile - nt * S) : 0;
#pragma unroll
        for (int rt = 0; rt < PJ_BM / 16; ++rt) {
            f4v acc0[3] = {zero4, zero4, zero4};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const h8v x1 = *(const h8v *)(&As[buf][0][rt * 16 + col][c * 32 + 8 * grp]);
                const h8v x2 = *(const h8v *)(&As[buf][1][rt * 16 + col][c * 32 + 8 * grp]);
                CTGCN_H2_MFMA1(Wf, c, x1, x2, acc0)
            }
            const int r_ = rt * 16 + col;
            const float rs = rscale[buf][r_];
            float *o;
            int64_t gstep;
            bool valid;
            if (S > 0) {
                valid = nt * PJ_BM + r_ < nodes;
                o = a.out + ((nt * S + tstep) * 3 * 8 + wave) * 1024 + r_ * 16 + 4 * grp;      // gi_blocked_offset(node, tstep, S, 0, oc)
                gstep = 8 * 1024;
            } else {
                const int64_t row = tile * PJ_BM + r_;
                valid = row < a.rows;
                o = a.out + row * (3 * GRU_H) + oc;
                gstep = GRU_H;
            }
            if (valid) {
#pragma unroll
                for (int g = 0; g < 3; ++g) *(f4v *)(o + g * gstep) = acc0[g] * (wsc[g] * rs) + bias[g];
            }
        }
        if (next < ntiles) stage_tile(buf ^ 1, stage);
        __syncthreads();
    }
}

// GRU recurrence, fp16x2 arithmetic: gru_seq_x3_kernel with two fp16 planes of h·2^14 (|h| < 1) instead of three bf16
// planes, W_hh rows scaled individually, 36 instead of 72 MFMAs per 16-row tile and step.  The scale of a product
// (row scale of W_hh × 2^-14) is folded into the gate pre-activation FMA, so the gate math costs what it did.
// bid / nblk: this block's index among the nblk blocks that share the work of `a` (the whole grid, or — grouped launch of a window's snapshots,
// gru_seq_h2_group_kernel — the blocks dealt to this snapshot)
template <bool REDUCE, bool SAVE, bool WIDE = false>        // WIDE: a row plan over 33-64 steps, two mask words per tile (its own instantiation)
__device__ __forceinline__ void gru_seq_h2_body(const GruArgs &a, const int bid, const int nblk)
{
    __shared__ _Float16 Hs[2][2][GRU_BM][PJ_PITCH];
    __shared__ float sbuf_[REDUCE ? 1 : 2][GRU_BM][GRU_PITCH];   // REDUCE: running sum over steps; otherwise fp32 h_t staged for the row-wise
    float(*const sbuf)[GRU_PITCH] = sbuf_[0];                     // LayerNorm / store, double buffered by step parity (emitted during the next step)
    __shared__ float wscale[4][GRU_H];             // rows 0-2: product scales of the three gates, row 3: b_hn
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int oc = wave * 16 + 4 * grp;       // first of the 4 consecutive hidden units this lane produces
    const int steps = a.steps;

    h8v Wf[2][4][3];      // [split][k chunk][gate]
    h2_load_weights<1>(a.whh, wave, col, grp, Wf, wscale);
    __syncthreads();
    // product scale per output column = row scale of W_hh x 2^-14 (the scale of h); kept in LDS together with b_hn and
    // re-read by every gate block (4 ds_read_b128) — the 16 VGPRs go to the deeper GI prefetch below
    if (grp == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) wscale[g][wave * 16 + col] *= (1.f / 16384.f);
        wscale[3][wave * 16 + col] = a.bhn ? a.bhn[wave * 16 + col] : 0.f;
    }
    __syncthreads();
    const int64_t ntiles = (a.rows + GRU_BM - 1) / GRU_BM;
    const int gstride = steps * 3 * GRU_H;

    for (int64_t tile_ = bid; tile_ < ntiles; tile_ += nblk) {
        const int64_t tile = ntiles - 1 - tile_;      // newest GI first: the projection kernel wrote the high tiles last
        const int64_t row0 = tile * GRU_BM;
        const int last = (int)min((int64_t)GRU_BM, a.rows - row0) - 1;
        // address of the lane's float4 of gate 0 for (step t, row tile rt); gates are gi_gs floats apart
        // = (scalar base of the tile) + one 32-bit lane offset: per-lane 64-bit pointers were spilled and reloaded inside the step loop, behind
        // the gi requests in flight (scratch is vector memory: the reload's wait is a wait for all of them)
        const int gi_gs = a.gi_blocked ? 8 * 1024 : GRU_H;
        // compact gi under a row plan (REDUCE form only): see GruArgs
        const bool planned = REDUCE && !SAVE && a.tmask != nullptr;
        typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type mask_t;
        mask_t tm = 0;
        if (planned) {
            if constexpr (WIDE) tm = (mask_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.tmask[2 * tile]) |
                                     ((mask_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.tmask[2 * tile + 1]) << (WIDE ? 32 : 0));
            else tm = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tmask[tile]);
        }
        auto popc = [](mask_t m) __attribute__((always_inline)) -> int { if constexpr (WIDE) return __popcll(m); else return __popc(m); };
        const int nfresh = popc(tm);
        const float *gi_base = planned ? a.gi + (int64_t)__builtin_amdgcn_readfirstlane(a.tbase[tile]) * (3 * GRU_H)
                                       : (a.gi_blocked ? a.gi + tile * (int64_t)gstride * GRU_BM : a.gi + row0 * gstride);
        const uint32_t gi_lane = a.gi_blocked ? (uint32_t)(wave * 1024 + col * 16 + 4 * grp) : (uint32_t)oc;
        auto gaddr = [&](int t, int rt) {
            uint32_t off;
            if (planned) off = (uint32_t)(min(rt * 16 + col, last) * nfresh + (popc(tm & (WIDE && t == 63 ? ~(mask_t)0 : (((mask_t)2 << t) - (mask_t)1))) - 1)) * (uint32_t)(3 * GRU_H);
            else off = a.gi_blocked ? (uint32_t)(t * (3 * 8 * 1024) + rt * 256) : (uint32_t)(t * 3 * GRU_H) + (uint32_t)min(rt * 16 + col, last) * (uint32_t)gstride;
            return gi_base + (gi_lane + off);
        };
        f4v hreg[GRU_RT];

        auto publish = [&](int buf, int r_, const f4v h) {     // split h·2^14 and store the two fp16 planes
            h4v p, q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 x, y;
                h2_split<1>(h[j] * 16384.f, x, y);
                p[j] = x; q[j] = y;
            }
            *(h4v *)(&Hs[buf][0][r_][oc]) = p;
            *(h4v *)(&Hs[buf][1][r_][oc]) = q;
        };
        // gate math for the lane's 4 hidden units of one row; ac: the accumulators of h_{t-1}·W_hh^T, still to be multiplied by csc
        auto gates = [&](const f4v gr, const f4v gz, const f4v gn, const f4v (&ac)[3], const f4v hold, int t, int r_) {     // t: the step of this h
            const f4v csc[3] = {*(const f4v *)(&wscale[0][oc]), *(const f4v *)(&wscale[1][oc]), *(const f4v *)(&wscale[2][oc])};
            const f4v b_hn = *(const f4v *)(&wscale[3][oc]);
            f4v h, rv, zv, nv, an;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rv[j] = gru_sigmoid(fmaf(ac[0][j], csc[0][j], gr[j]));
                zv[j] = gru_sigmoid(fmaf(ac[1][j], csc[1][j], gz[j]));
                an[j] = fmaf(ac[2][j], csc[2][j], b_hn[j]);
                nv[j] = gru_tanh(fmaf(rv[j], an[j], gn[j]));
                h[j] = nv[j] + zv[j] * (hold[j] - nv[j]);
            }
            if (!REDUCE) *(f4v *)(&sbuf_[REDUCE ? 0 : (t & 1)][r_][oc]) = h;
            if (SAVE && r_ <= last) {
                float *gp = a.gates + ((row0 + r_) * steps + t) * (4 * GRU_H) + oc;
                *(f4v *)gp = rv; *(f4v *)(gp + GRU_H) = zv; *(f4v *)(gp + 2 * GRU_H) = nv; *(f4v *)(gp + 3 * GRU_H) = an;
            }
            return h;
        };
        const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};
        const f4v zero3[3] = {zero4, zero4, zero4};

        // ---- step 0: h_{-1} = 0, no MFMA
#pragma unroll
        for (int rt = 0; rt < GRU_RT; ++rt) {
            const float *p = gaddr(0, rt);
            const f4v h = gates(*(const f4v *)p, *(const f4v *)(p + gi_gs), *(const f4v *)(p + 2 * gi_gs), zero3, zero4, 0, rt * 16 + col);
            hreg[rt] = h;
            publish(0, rt * 16 + col, h);
            if (REDUCE) *(f4v *)(&sbuf[rt * 16 + col][oc]) = h;
        }
        __syncthreads();
        // per-step output: LayerNorm (or plain copy) of the staged fp32 rows of step t, 512 B per row.  No barrier of its own: it runs
        // behind the first barrier of step t + 1 (all 64 rows of step t are staged by then), and the buffer is next written by the
        // gate math of step t + 2, two barriers later
        auto emit_step = [&](int t) {
            for (int r = wave; r <= last; r += 8)
                gru_layernorm_row(sbuf_[REDUCE ? 0 : (t & 1)][r], a.out + ((row0 + r) * steps + t) * GRU_H, lane, a.gamma, a.beta, a.eps);
        };
        // Software pipeline over the row tiles of a step: the gate math of tile rt-1 is issued among tile rt's MFMAs, and
        // the GI operands of a row tile are requested TWO units (row tile x step) before its MFMAs start — a full step
        // before its gate math — into four statically indexed buffers (the loaded HBM latency exceeds one step's MFMAs).
        f4v gq[4][3];
        auto load_gi = [&](int t, int rt) {
            const float *p = gaddr(t, rt);
            gq[rt][0] = *(const f4v *)p; gq[rt][1] = *(const f4v *)(p + gi_gs); gq[rt][2] = *(const f4v *)(p + 2 * gi_gs);
        };
        if (steps > 1) { load_gi(1, 0); load_gi(1, 1); }
        auto mfma_unit = [&](int pb, int rt, f4v (&ac)[3]) {
#pragma unroll
            for (int g = 0; g < 3; ++g) ac[g] = zero4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // MFMA B operand: B[k][n = row] = h_{t-1}[row rt*16 + col][k = c*32 + 8*grp + j]
                const h8v x1 = *(const h8v *)(&Hs[pb][0][rt * 16 + col][c * 32 + 8 * grp]);
                const h8v x2 = *(const h8v *)(&Hs[pb][1][rt * 16 + col][c * 32 + 8 * grp]);
                CTGCN_H2_MFMA1(Wf, c, x1, x2, ac)
            }
        };
        if constexpr (REDUCE) {
            // Rolling pipeline ACROSS steps: unit (t, rt) issues its MFMAs together with the gate math of the previous unit —
            // for rt = 0 that is row tile 3 of step t-1, so the matrix pipe does not drain at step boundaries.  Row tiles
            // are independent sequences: h_{t-1}[rows of rt] is complete three units before MFMA(t, rt) needs it, and
            // two barriers per step (after rt = 1 and rt = 3) always fall in between.  The publish of (t-1, 3) goes to the
            // buffer step t reads, rows 3 — not yet read by anyone in step t.
            f4v acc[2][3];
            for (int t = 1; t < steps; ++t) {
                const int pb = (t - 1) & 1, cb = t & 1;
#pragma unroll
                for (int rt = 0; rt < GRU_RT; ++rt) {
                    const int cur = rt & 1, prv = cur ^ 1;
                    if (rt + 2 < GRU_RT) load_gi(t, rt + 2);
                    else if (t + 1 < steps) load_gi(t + 1, rt + 2 - GRU_RT);
                    mfma_unit(pb, rt, acc[cur]);
                    if (rt > 0 || t > 1) {
                        const int rp = rt > 0 ? rt - 1 : GRU_RT - 1;          // the previous unit's row tile ...
                        const int wb = rt > 0 ? cb : pb;                      // ... and the plane buffer of ITS step
                        const f4v h = gates(gq[rp][0], gq[rp][1], gq[rp][2], acc[prv], hreg[rp], t, rp * 16 + col);
                        hreg[rp] = h;
                        publish(wb, rp * 16 + col, h);
                        f4v *sp_ = (f4v *)(&sbuf[rp * 16 + col][oc]);
                        *sp_ = *sp_ + h;
                    }
                    if (rt & 1) __syncthreads();
                }
            }
            if (steps > 1) {                                                  // drain: row tile 3 of the last step
                const int rp = GRU_RT - 1;
                const f4v h = gates(gq[rp][0], gq[rp][1], gq[rp][2], acc[1], hreg[rp], steps - 1, rp * 16 + col);
                f4v *sp_ = (f4v *)(&sbuf[rp * 16 + col][oc]);
                *sp_ = *sp_ + h;
            }
            __syncthreads();
        } else {
            // The same rolling pipeline for the per-step form (temporal GRU: LayerNorm(h_t) of every step is an output).  It used to
            // drain at every step: all four gate blocks, barrier, LayerNorm of 64 rows, barrier — 9.5 ms per 1 M x 16 call against an
            // HBM bound of 5.8.  Now the rows of step t-1 leave while step t multiplies.
            f4v acc[2][3];
            for (int t = 1; t < steps; ++t) {
                const int pb = (t - 1) & 1, cb = t & 1;
#pragma unroll
                for (int rt = 0; rt < GRU_RT; ++rt) {
                    const int cur = rt & 1, prv = cur ^ 1;
                    if (rt + 2 < GRU_RT) load_gi(t, rt + 2);
                    else if (t + 1 < steps) load_gi(t + 1, rt + 2 - GRU_RT);
                    mfma_unit(pb, rt, acc[cur]);
                    if (rt > 0 || t > 1) {
                        const int rp = rt > 0 ? rt - 1 : GRU_RT - 1;          // the previous unit's row tile ...
                        const int wb = rt > 0 ? cb : pb;                      // ... the plane buffer of ITS step ...
                        const int ts = rt > 0 ? t : t - 1;                    // ... and that step
                        const f4v h = gates(gq[rp][0], gq[rp][1], gq[rp][2], acc[prv], hreg[rp], ts, rp * 16 + col);
                        hreg[rp] = h;
                        publish(wb, rp * 16 + col, h);
                    }
                    if (rt & 1) __syncthreads();
                    if (rt == 1) emit_step(t - 1);                            // its last row tile finished in unit (t, 0), before that barrier
                }
            }
            if (steps > 1) {                                                  // drain: row tile 3 of the last step
                const int rp = GRU_RT - 1;
                gates(gq[rp][0], gq[rp][1], gq[rp][2], acc[1], hreg[rp], steps - 1, rp * 16 + col);
            }
            __syncthreads();
            emit_step(steps - 1);
        }
        if (REDUCE)
            for (int r = wave; r <= last; r += 8) {
                const int64_t orow = (planned && a.order) ? (int64_t)a.order[row0 + r] : row0 + r;
                gru_layernorm_row(sbuf[r], a.out + orow * a.ldo, lane, a.gamma, a.beta, a.eps);
            }
        __syncthreads();       // LDS is reused by the next tile
    }
}

template <bool REDUCE, bool SAVE, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void gru_seq_h2_kernel(const GruArgs a)
{
    gru_seq_h2_body<REDUCE, SAVE, WIDE>(a, (int)blockIdx.x, (int)gridDim.x);
}

// the recurrences of a small window's snapshots in one launch (round 5: the 500-wide first layer): block -> (snapshot, index among its blocks,
// their number) from the block map, the snapshot's arguments (its own W_hh, LayerNorm, gi, row plan) from the table; the body is the kernel's
__global__ __launch_bounds__(512, 2) void gru_seq_h2_group_kernel(const GruArgs *__restrict__ table, const int32_t *__restrict__ blockmap)
{
    const int g = __builtin_amdgcn_readfirstlane(blockmap[3 * blockIdx.x]);
    const int bid = __builtin_amdgcn_readfirstlane(blockmap[3 * blockIdx.x + 1]);
    const int nblk = __builtin_amdgcn_readfirstlane(blockmap[3 * blockIdx.x + 2]);
    const GruArgs a = table[g];
    gru_seq_h2_body<true, false>(a, bid, nblk);
}

// ------------------------------------------------------------------------------------------------
// gru_layer_h2_kernel: input projection AND recurrence of a node tile in one kernel with BOTH weight matrices resident in
// the register file — GI is never materialised anywhere (not in HBM, not in the memory-side cache, not in LDS).
// Why it fits: W_ih and W_hh as fp16x2 MFMA fragments are 2 x 196 KB = 77 % of a CU's 512 KB unified VGPR/AGPR file.  A
// block is therefore FOUR waves, one per SIMD (512 registers each): wave w owns hidden units [32w, 32w + 32) of all three
// gates of both matrices = 2 x 192 registers of ready-made MFMA A operands, most of them in the accumulator half of the
// file (MFMA reads A/B operands from AGPRs directly), which leaves the 128-odd architectural VGPRs a wave needs for
// accumulators, B operands and gate math.  Per (step, 16-row tile) a wave issues 72 MFMAs for x_t·W_ihᵀ and 72 for
// h_{t-1}·W_hhᵀ and finishes its 32 hidden units' gate math straight from the accumulators.
// Data flow per block (persistent, 64-row tiles, "unit" = one step of one 16-row sub-tile):
//   global x -> registers (two units ahead) -> per-row power-of-two scale + fp16x2 split -> LDS ring of two 16-row slots;
//   h_t: fp32 in LDS (the lane that wrote a value is the only one that reads it back) + fp16x2 planes in LDS for the
//   next step's MFMAs, published one unit late so that one barrier per unit orders every hazard;
//   REDUCE: running sum in LDS, LayerNorm at the end of the tile; otherwise LayerNorm(h_t) rows are emitted per unit.
// Arithmetic is that of gru_proj_h2_kernel + gru_seq_h2_kernel operation for operation (same splits, same MFMA order
// per accumulator, same epilogue expressions): results are bit-identical to the kernel pair.
// HBM traffic: x in (512 B per row-step) + out; the pair moves 3.6 KB per row-step.
// (Round 1's attempt kept the pair's 8-wave blocks, time-shared the weight registers between a projection phase and a recurrence
// phase of the same tile and parked GI in a per-block scratch served by the memory-side cache: bit-identical, but 15 % SLOWER
// than the pair — the round trip costs Infinity-Fabric bandwidth whether it ends in HBM or in the cache.  Removed.)
// ------------------------------------------------------------------------------------------------
constexpr int LY_BM = 32;     // rows per tile (2 MFMA row tiles): small, so that LDS has room for weight fragments
constexpr int LY_RT = LY_BM / 16;
constexpr int LY_WL_TOTAL = 80;   // W_ih fragments of the block kept in LDS (1 KB each): 80 KB, split evenly over the waves
// Fragment (unit tile of the wave, split, k chunk, gate) of W_ih that lives in LDS, or -1.  Residual plane first.
//   NW = 4 (two unit tiles per wave, 20 slots): chunks 0-2 of both unit tiles + (chunk 3, ut 0, gates r z)
//   NW = 8 (one unit tile per wave, 10 slots):  chunks 0-2 + (chunk 3, gate r)
template <int NW>
__device__ __forceinline__ constexpr int ly_lds_slot(int ut, int sp, int c, int g)
{
    if (NW == 4) return (sp == 1 && c < 3) ? (c * 2 + ut) * 3 + g : ((sp == 1 && c == 3 && ut == 0 && g < 2) ? 18 + g : -1);
    return (sp == 1 && c < 3) ? c * 3 + g : ((sp == 1 && c == 3 && g == 0) ? 9 : -1);
}
// ... and these W_ih fragments are pinned to AGPRs next to all of W_hh; the rest of the AGPRs is left to the MFMA accumulators.
template <int NW>
__device__ __forceinline__ constexpr bool ly_in_agpr(int ut, int sp, int c, int g)
{
    if (NW == 4) return (sp == 1 && ly_lds_slot<4>(ut, sp, c, g) < 0) || (sp == 0 && c >= 2);                  // 48 + 16 fragments = 256 registers
    return (sp == 1 && ly_lds_slot<8>(ut, sp, c, g) < 0) || (sp == 0 && c >= 2);                               // 24 + 8 fragments = 128 registers
}

struct LayerArgs {
    int64_t rows;            // sequences (nodes)
    int32_t steps;
    const float *x;          // [rows, steps, 128], row-step stride ldx
    int64_t ldx;
    const float *wih, *whh;  // [384, 128] each
    const float *bias_gi;    // [384] or null: b_ih (+ b_hh for r, z)
    const float *bhn;        // [128] or null
    const float *gamma, *beta;
    float eps;
    float *out;              // REDUCE: [rows, ldo]; else [rows, steps, 128]
    int64_t ldo;
    const _Float16 *xp1, *xp2;   // PRESPLIT (gru_layer8_h2_kernel<true>): the fp16 planes [rows * steps, 128] and row scales of x,
    const float *xps;            // as ctgcn_core_aggregate_split_f32 writes them (x is unused then)
    // PRESPLIT + REDUCE: the plan ctgcn_core_aggregate_split_f32 wrote the planes under (both null: natural order, every step present).
    // order[p] = output row of sequence p; bit t of tmask[tile] clear = x_t of all 16 sequences of the tile repeats x_{t-1} (not stored):
    // the x·W_ih products of step t-1 are kept, the step costs the h·W_hh half only.  Bit 0 is always set; steps <= 32.
    const int32_t *order;
    const uint32_t *tmask;
    // x of step t of sequence r at x + r ld_row + step_off[t] instead of x + (r steps + t) ldx (both null / 0: the dense layout): the temporal
    // GRU of a snapshot-parallel forward reads the all-to-all's receive buffer [slot][rank][node][128] in time order without a copy
    const int64_t *step_off;
    int64_t ld_row;
    float *gates;                // SAVE (per-step form without LayerNorm): [rows, steps, 4, 128] r, z, n, q = W_hn h + b_hn for ctgcn_gru_seq_bwd_f32
    // SAVE on the planes + row plan form (training's recompute pass under the plan, ctgcn_gru_layer_presplit_save_f32): gates as above, the raw
    // h sequence [rows, steps, 128] and the sum over the steps BEFORE the LayerNorm [rows, 128] (what its backward needs) — all in POSITION
    // order (row p of these buffers is sequence p of the planes; `order` is not applied, `out` is not written)
    float *hseq, *presum;
    int32_t gates3;              // SAVE: gates are [rows, steps, 3, 128] = r, z, q — n is rebuilt by ctgcn_gru_bwd_rec_f32 from h_t, h_{t-1} and z (0.5 KB per row-step less, written and read)
#ifdef CTGCN_LAYER_TIMELINE
    unsigned long long *timeline;   // diagnostic build: per (block, wave) sums of the unit phases, see tools/layer_timeline.py
#endif
};

// weight fragments of ONE 16-unit tile (tile16 = hidden units [16*tile16, 16*tile16+16)) of all three gates, as h2_load_weights
#ifdef CTGCN_LAYER_TIMELINE
const char *timeline_begin(LayerArgs &a, unsigned nb8, void *stream)
{
    const char *tl_file = getenv("CTGCN_LAYER_TIMELINE_FILE");
    a.timeline = nullptr;
    if (tl_file) { (void)hipMalloc(&a.timeline, (size_t)nb8 * 8 * 12 * 8); (void)hipMemsetAsync(a.timeline, 0, (size_t)nb8 * 8 * 12 * 8, (hipStream_t)stream); }
    return tl_file;
}
void timeline_end(LayerArgs &a, unsigned nb8, const char *tl_file, void *stream)
{
    if (!a.timeline) return;
    (void)hipStreamSynchronize((hipStream_t)stream);
    unsigned long long *h = (unsigned long long *)malloc((size_t)nb8 * 8 * 12 * 8);
    (void)hipMemcpy(h, a.timeline, (size_t)nb8 * 8 * 12 * 8, hipMemcpyDeviceToHost);
    FILE *f = fopen(tl_file, "w");          // rewritten by every call: the last call's numbers stay
    for (unsigned i = 0; i < nb8 * 8; ++i) {
        fprintf(f, "%u %u", i / 8, i % 8);
        for (int j = 0; j < 12; ++j) fprintf(f, " %llu", h[i * 12 + j]);
        fprintf(f, "\n");
    }
    fclose(f);
    free(h);
    (void)hipFree(a.timeline);
    a.timeline = nullptr;
}
#endif
template <int RS>
__device__ __forceinline__ void h2_load_weight_tile(const float *w, int tile16, int col, int grp, h8v (&Wf)[2][4][3], float (*wscale)[GRU_H])
{
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float *row = w + (int64_t)(g * GRU_H + tile16 * 16 + col) * GRU_H;
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4v lo = *(const f4v *)(row + c * 32 + 8 * grp), hi = *(const f4v *)(row + c * 32 + 8 * grp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) m = fmaxf(m, fmaxf(fabsf(lo[j]), fabsf(hi[j])));
        }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        float sc, inv;
        h2_scale(m, sc, inv);
        if (grp == 0) wscale[g][tile16 * 16 + col] = sc;
#pragma unroll
        for (int c = 0; c < 4; ++c) h2_split_x8<RS>(row + c * 32 + 8 * grp, inv, Wf[0][c][g], Wf[1][c][g]);
    }
}

// NW = waves per block: 4 (one per SIMD, 512 registers each, two 16-unit tiles per wave) or 8 (two per SIMD, 256 registers
// each, one 16-unit tile per wave: the second wave of a SIMD covers the first one's LDS latency).
template <bool REDUCE, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void gru_layer_h2_kernel(const LayerArgs a)
{
    constexpr int UTW = 8 / NW;                          // 16-unit tiles per wave
    constexpr int NT = 64 * NW;
    constexpr int WL = LY_WL_TOTAL / NW;                 // LDS-resident W_ih fragments per wave
    constexpr int SL = 4 * NW;                           // staging lanes per x row (16 rows per unit)
    constexpr int SF = GRU_H / SL;                       // floats per staging lane: 8 or 4
    __shared__ _Float16 Xs[2][2][16][PJ_PITCH];          // ring of two units: the two fp16 planes of 16 rows of x_t
    __shared__ _Float16 Hs[2][LY_BM][PJ_PITCH];          // the two fp16 planes of h_{t-1}·2^14 for the tile's rows
    __shared__ float hold[LY_BM][GRU_PITCH];             // h_{t-1} in fp32 (each value is read back only by the lane that wrote it)
    __shared__ float hsum[LY_BM][GRU_PITCH];             // REDUCE: running sum over steps
    __shared__ float xscale[2][16];
    __shared__ float wsc_ih[3][GRU_H];
    __shared__ float csc_hh[4][GRU_H];                   // rows 0-2: product scales of the three gates, row 3: b_hn
    __shared__ float bias_s[3][GRU_H];
    __shared__ h8v Wl[NW][WL][64];                       // per wave: the W_ih fragments that do not fit the register file
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int S = a.steps;

    h8v Wi[UTW][2][4][3], Wh[UTW][2][4][3];              // [unit tile][split][k chunk][gate]
#pragma unroll
    for (int ut = 0; ut < UTW; ++ut) {
        h2_load_weight_tile<1>(a.wih, wave * UTW + ut, col, grp, Wi[ut], wsc_ih);
        h2_load_weight_tile<1>(a.whh, wave * UTW + ut, col, grp, Wh[ut], csc_hh);
    }
    // Register-file placement: MFMA A operands may live in the accumulator half (AGPRs) of the unified file, everything the
    // VALU touches must be in the architectural half.  The empty asm gives a value the AGPR register class for its whole
    // live range (the MFMAs read it there); WL fragments per wave go to LDS and are read on the spot; the rest stays in VGPRs.
#pragma unroll
    for (int ut = 0; ut < UTW; ++ut)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    asm volatile("" : "+a"(Wh[ut][sp][c][g]));
                    if (ly_in_agpr<NW>(ut, sp, c, g)) asm volatile("" : "+a"(Wi[ut][sp][c][g]));
                    if (ly_lds_slot<NW>(ut, sp, c, g) >= 0) Wl[wave][ly_lds_slot<NW>(ut, sp, c, g)][lane] = Wi[ut][sp][c][g];
                }
    __syncthreads();
    for (int i = tid; i < 3 * GRU_H; i += NT) {
        (&csc_hh[0][0])[i] *= (1.f / 16384.f);           // x the scale of h (planes hold h·2^14)
        (&bias_s[0][0])[i] = a.bias_gi ? a.bias_gi[i] : 0.f;
    }
    if (tid < GRU_H) csc_hh[3][tid] = a.bhn ? a.bhn[tid] : 0.f;
    __syncthreads();

    const int64_t ntiles = (a.rows + LY_BM - 1) / LY_BM;
    const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};
    // staging role: 16 rows x SL lanes, a lane holds SF consecutive floats of its row
    const int sr = tid / SL, sc = (tid % SL) * SF;
    struct Unit { int64_t tile; int t, rt; };
    auto advance = [&](Unit u, int n) {                  // n units later in this block's (tile, step, row tile) order
        u.rt += n;
        u.t += u.rt / LY_RT; u.rt %= LY_RT;
        while (u.t >= S) { u.t -= S; u.tile += gridDim.x; }
        return u;
    };
    auto load_x = [&](const Unit u, f4v (&v)[SF / 4]) {
        if (u.tile < ntiles) {
            const int64_t row = min(u.tile * LY_BM + u.rt * 16 + sr, a.rows - 1);
            const float *p = a.x + (row * S + u.t) * a.ldx + sc;
#pragma unroll
            for (int i = 0; i < SF / 4; ++i) v[i] = *(const f4v *)(p + 4 * i);
        }
    };
    auto stage_x = [&](int slot, const f4v (&v)[SF / 4]) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < SF / 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(v[i][j]));
#pragma unroll
        for (int d = 1; d < SL; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
        float scl, inv;
        h2_scale(m, scl, inv);
        if ((tid % SL) == 0) xscale[slot][sr] = scl;
#pragma unroll
        for (int i = 0; i < SF / 4; ++i) {
            h4v s0, s1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 p, q;
                h2_split<1>(v[i][j] * inv, p, q); s0[j] = p; s1[j] = q;
            }
            *(h4v *)(&Xs[slot][0][sr][sc + 4 * i]) = s0;
            *(h4v *)(&Xs[slot][1][sr][sc + 4 * i]) = s1;
        }
    };

    Unit cur{(int64_t)blockIdx.x, 0, 0};
    if (cur.tile >= ntiles) return;
    f4v xr[SF / 4];                                       // x of the NEXT unit, requested one unit (~2.7 us) before it is staged
    load_x(cur, xr);
    stage_x(0, xr);
    load_x(advance(cur, 1), xr);
    __syncthreads();

    f4v hpub[UTW];                                        // h of the previous unit, published one unit late
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * LY_BM;
        const int last = (int)min((int64_t)LY_BM, a.rows - row0) - 1;
        for (int t = 0; t < S; ++t) {
#pragma unroll
            for (int rt = 0; rt < LY_RT; ++rt) {
                const int slot = rt & 1;
                // ---- the previous unit's leftovers (its rows are complete: the barrier that ended it has been passed)
                if (rt > 0 || t > 0) {
                    const int rp = (rt + LY_RT - 1) & (LY_RT - 1);
                    const int r_ = rp * 16 + col;
#pragma unroll
                    for (int ut = 0; ut < UTW; ++ut) {   // fp16x2 planes of h·2^14 for the next step's MFMAs
                        const int oc = (wave * UTW + ut) * 16 + 4 * grp;
                        h4v p, q;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            _Float16 x, y;
                            h2_split<1>(hpub[ut][j] * 16384.f, x, y);
                            p[j] = x; q[j] = y;
                        }
                        *(h4v *)(&Hs[0][r_][oc]) = p;
                        *(h4v *)(&Hs[1][r_][oc]) = q;
                    }
                    if (!REDUCE) {                       // LayerNorm(h_t) of the previous unit's 16 rows, 16 / NW rows per wave
                        const int tp = rt > 0 ? t : t - 1;
                        for (int r = rp * 16 + wave * (16 / NW); r < rp * 16 + (wave + 1) * (16 / NW); ++r)
                            if (r <= last) gru_layernorm_row(hold[r], a.out + ((row0 + r) * S + tp) * GRU_H, lane, a.gamma, a.beta, a.eps);
                    }
                }
                // ---- x of the next unit: registers -> planes; request the unit after the next two
                {
                    const Unit nx = advance(Unit{tile, t, rt}, 1);
                    if (nx.tile < ntiles) stage_x(slot ^ 1, xr);
                    load_x(advance(Unit{tile, t, rt}, 2), xr);
                }
                // ---- this unit: the wave's UTW x 16 hidden units (unit-tile-major measured fastest: 6.16 ms per 1M x 8 call, against
                // 6.40 with the B operands of a chunk shared by both unit tiles and 6.84 with explicitly double-buffered operand
                // sets — both cost VGPRs the allocator then takes back as scratch reloads inside the MFMA stream)
                const int r_ = rt * 16 + col;
                const float rs = xscale[slot][col];
#pragma unroll
                for (int ut = 0; ut < UTW; ++ut) {
                    f4v acc0[3] = {zero4, zero4, zero4}, ach[3] = {zero4, zero4, zero4};
                    // Issue order is pinned with sched_barriers (left alone, the scheduler sinks every ds_read to just before its
                    // first use and waits with lgkmcnt(0): one exposed LDS latency per three MFMAs, and a single wave per SIMD has
                    // nobody to cover it).  Per k chunk: the h planes and the LDS-resident weight fragments are requested before
                    // the nine x MFMAs, the x planes of the NEXT chunk before the nine h MFMAs.
                    auto body = [&](auto with_h_tag) {
                        constexpr bool with_h = decltype(with_h_tag)::value;
                        // (hoisting these first loads above the previous unit's publish / staging work, or above the previous unit
                        // tile's gate math, was measured slower: 5.93 vs 5.43 ms — the extra live registers come back as spills)
                        h8v xa1 = *(const h8v *)(&Xs[slot][0][col][8 * grp]);
                        h8v xa2 = *(const h8v *)(&Xs[slot][1][col][8 * grp]);
                        h8v wl[3];
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const int sl = ly_lds_slot<NW>(ut, 1, 0, g);
                            if (sl >= 0) wl[g] = Wl[wave][sl < 0 ? 0 : sl][lane];
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            h8v h1, h2;
                            if (with_h) {
                                h1 = *(const h8v *)(&Hs[0][r_][c * 32 + 8 * grp]);
                                h2 = *(const h8v *)(&Hs[1][r_][c * 32 + 8 * grp]);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            // = CTGCN_H2_MFMA1(Wi[ut], c, x1, x2, acc0)
#pragma unroll
                            for (int g = 0; g < 3; ++g) acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wi[ut][0][c][g], xa2, acc0[g], 0, 0, 0);
#pragma unroll
                            for (int g = 0; g < 3; ++g)
                                acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ly_lds_slot<NW>(ut, 1, c, g) >= 0 ? wl[g] : Wi[ut][1][c][g], xa1, acc0[g], 0, 0, 0);
#pragma unroll
                            for (int g = 0; g < 3; ++g) acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wi[ut][0][c][g], xa1, acc0[g], 0, 0, 0);
                            if (c < 3) {
                                __builtin_amdgcn_sched_barrier(0);
                                xa1 = *(const h8v *)(&Xs[slot][0][col][(c + 1) * 32 + 8 * grp]);
                                xa2 = *(const h8v *)(&Xs[slot][1][col][(c + 1) * 32 + 8 * grp]);
#pragma unroll
                                for (int g = 0; g < 3; ++g) {
                                    const int sl = ly_lds_slot<NW>(ut, 1, c + 1, g);
                                    if (sl >= 0) wl[g] = Wl[wave][sl < 0 ? 0 : sl][lane];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (with_h) { CTGCN_H2_MFMA1(Wh[ut], c, h1, h2, ach) }
                        }
                    };
                    if (t > 0) body(std::true_type{}); else body(std::false_type{});
                    const int oc = (wave * UTW + ut) * 16 + 4 * grp;
                    f4v gi[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        gi[g] = acc0[g] * (*(const f4v *)(&wsc_ih[g][oc]) * rs) + *(const f4v *)(&bias_s[g][oc]);
                    const f4v csc[3] = {*(const f4v *)(&csc_hh[0][oc]), *(const f4v *)(&csc_hh[1][oc]), *(const f4v *)(&csc_hh[2][oc])};
                    const f4v b_hn = *(const f4v *)(&csc_hh[3][oc]);
                    const f4v hprev = t > 0 ? *(const f4v *)(&hold[r_][oc]) : zero4;
                    f4v h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float rv = gru_sigmoid(fmaf(ach[0][j], csc[0][j], gi[0][j]));
                        const float zv = gru_sigmoid(fmaf(ach[1][j], csc[1][j], gi[1][j]));
                        const float an = fmaf(ach[2][j], csc[2][j], b_hn[j]);
                        const float nv = gru_tanh(fmaf(rv, an, gi[2][j]));
                        h[j] = nv + zv * (hprev[j] - nv);
                    }
                    *(f4v *)(&hold[r_][oc]) = h;
                    if (REDUCE) {
                        f4v *sp_ = (f4v *)(&hsum[r_][oc]);
                        *sp_ = t > 0 ? *sp_ + h : h;
                    }
                    hpub[ut] = h;
                }
                __syncthreads();
            }
        }
        // ---- end of the tile
        if (REDUCE) {
            for (int r = wave * (LY_BM / NW); r < (wave + 1) * (LY_BM / NW); ++r)
                if (r <= last) gru_layernorm_row(hsum[r], a.out + (row0 + r) * a.ldo, lane, a.gamma, a.beta, a.eps);
        } else {
            for (int r = (LY_RT - 1) * 16 + wave * (16 / NW); r < (LY_RT - 1) * 16 + (wave + 1) * (16 / NW); ++r)
                if (r <= last) gru_layernorm_row(hold[r], a.out + ((row0 + r) * S + (S - 1)) * GRU_H, lane, a.gamma, a.beta, a.eps);
        }
        __syncthreads();       // hold / hsum rows are rewritten by the next tile's first units
    }
}

// ------------------------------------------------------------------------------------------------
// gru_layer8_h2_kernel: the same fusion with EIGHT waves per block — two per SIMD, so that one wave's LDS / MFMA-result
// latency is covered by the other instead of by hand-placed loads.  256 registers per wave; wave w owns hidden units
// [16w, 16w+16) of both matrices = 48 fragments: W_hh (24) + 8 of W_ih pinned to AGPRs, the 12 fragments of W_ih's residual plane in
// LDS (96 KB), 4 in VGPRs (15 in LDS until the x products went to one accumulator and freed 12 registers: 4.36 -> 4.28 ms; with 9
// in LDS the allocator starts reloading from scratch inside the unit: 4.33).  That much LDS for weights forces 16-row tiles: one unit per step,
// h planes double buffered by step parity (published right after the gate math), a lane owns ONE (row, 4 hidden units) patch
// for the whole tile, so h_{t-1} and the running sum stay in registers; the sum goes through the idle x slot for the final
// LayerNorm.  Sum-over-steps form only.  Same arithmetic as the other two paths (bit-identical).
// ------------------------------------------------------------------------------------------------
#ifndef CTGCN_LAYER_LOOKAHEAD
#define CTGCN_LAYER_LOOKAHEAD 1      // 0: A/B build without the early x products of gru_layer8_h2_kernel<presplit, sum>
#endif
#ifndef CTGCN_X_GLDS
#define CTGCN_X_GLDS 0               // 1: A/B build — the row-plan path's x planes travel global -> LDS directly (global_load_lds), ring of three slots.
                                     // Measured in round 6 (tools/runs/r6_glds_ab.sh): bit-identical, 3.56 / 4.25 ms against 3.23 / 3.83 ms per 1 M x 8 call
                                     // (snapshots 3 / 15): 10 % SLOWER — hipcc drains the DMA queue (vmcnt(0)) in front of every s_barrier, one per unit
#endif
constexpr int L8_WL = 12;
constexpr int L8_PITCH = 128;                            // halfs per plane row, no padding
// half index of element k of row r (r < 16): 16-byte segment k / 8 goes to segment (k / 8) ^ r
__device__ __forceinline__ int l8_off(int r, int k) { return ((((k >> 3) ^ r) & 15) << 3) | (k & 7); }
__device__ __forceinline__ constexpr int l8_lds_slot(int sp, int c, int g) { return sp == 1 ? c * 3 + g : -1; }
// WL fragments of W_ih in LDS: the 12 of the residual plane, then the LAST WL - 12 of the leading plane
template <int WL>
__device__ __forceinline__ constexpr int l8_slot(int sp, int c, int g) { return sp == 1 ? c * 3 + g : (c * 3 + g >= 24 - WL ? c * 3 + g - (24 - WL) + 12 : -1); }

// bid / nblk: this block's index among the nblk blocks that share the work of `a` (the whole grid, or — grouped launch of a window's
// snapshots, gru_layer8_h2_group_kernel — the blocks assigned to this snapshot)
// WIDE (round 5): a row plan over up to 64 steps (America-Air max core 64, Europe-Air 33): two mask words per tile, tmask[2 tile] = steps
// 0-31, tmask[2 tile + 1] = steps 32-63.  A separate instantiation: the K <= 32 kernels keep their 32-bit masks and their code.
template <bool PRESPLIT, bool REDUCE, bool SAVE, bool WIDE = false>
__device__ __forceinline__ void gru_layer8_h2_body(const LayerArgs &a, const int bid, const int nblk)
{
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type mask_t;
    constexpr mask_t MASK_ALL = ~(mask_t)0;
    // bit t of a step mask.  Without a plan the mask is all ones and t runs to steps - 1 (up to 254): the shift count is reduced to the mask's
    // width (a C++ shift by >= the width is undefined — round 4's plan-less kernel at 40 steps gave rows off by 1e-1, found in round 5 by
    // the first comparison of that path with the fp32-H path at K > 32)
    constexpr int MASK_BITS = WIDE ? 63 : 31;
    auto mask_bit = [](mask_t m, int t) __attribute__((always_inline)) -> bool { return ((m >> (t & MASK_BITS)) & 1) != 0; };
    auto uniform_mask = [](mask_t m) __attribute__((always_inline)) -> mask_t {
        if constexpr (WIDE) return (mask_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m) |
                                   ((mask_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)m >> 32)) << (WIDE ? 32 : 0));
        else return (mask_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    };
    // x and h planes: 16 rows of 128 halfs, unpadded; the 16-byte segment q of row r is stored at segment q ^ r (l8_off).  Every access
    // pattern of the kernel is then conflict-free: the MFMA operand reads (ds_read_b128: lane = (row, k group) — with the 8-half row
    // padding of the other GRU kernels rows 11 and 12 met in one bank group: SQ_LDS_BANK_CONFLICT was 26 % of the LDS cycles), the
    // staging writes (32 lanes along a row) and the publish writes (16 rows at one column block).  It also frees 2 KB of LDS.
    constexpr bool GLDS = CTGCN_X_GLDS && CTGCN_LAYER_LOOKAHEAD && PRESPLIT && REDUCE && !SAVE;
    constexpr int XNS = GLDS ? 3 : 2;
    __shared__ _Float16 Xs[XNS][2][16][L8_PITCH];        // ring of two (GLDS build: three) units: the two fp16 planes of 16 rows of x_t
    __shared__ _Float16 Hs[2][2][16][L8_PITCH];          // [step parity][plane]: h_t·2^14 (after the last step: the summed rows in fp32)
    __shared__ float xscale[2][16];
    __shared__ float wsc_ih[3][GRU_H];
    __shared__ float csc_hh[4][GRU_H];                   // rows 0-2: product scales of the three gates, row 3: b_hn
    __shared__ float bias_s[3][GRU_H];
    __shared__ float ln_gb[2][GRU_H];                    // LayerNorm weight / bias and the temporal layout's step offsets: no vector-memory load
    __shared__ int64_t soff_s[32];                       // outside the x pipeline (any wait on one is a wait for the x rows in flight, see below)
    __shared__ int32_t ord_s[4][16];                     // row-plan forms: output rows and step mask of the tiles in flight (ring of four, see below)
    __shared__ mask_t msk_s[4];
    // the row-plan forms have no hrow staging: room for more fragments.  The recompute pass (SAVE) takes two more: with 12 the x prefetch
    // registers were spilled right behind their loads (s_waitcnt + scratch_store: the prefetch distance became zero)
    constexpr int WL = (PRESPLIT && REDUCE) ? (SAVE ? 15 : 12) : L8_WL;
    __shared__ h8v Wl[8][WL][64];
    // per-step form: h_t of the unit's 16 rows in fp32, double buffered by unit parity; the rows leave (LayerNorm, 512-byte stores) at the
    // start of the NEXT unit, two per wave — the staging the kernel pair uses, so the outputs are the pair's bit for bit
    __shared__ float hrow[REDUCE ? 1 : 2][REDUCE ? 1 : 16][GRU_PITCH];
    static_assert(sizeof(_Float16) * 2 * 16 * L8_PITCH >= sizeof(float) * 16 * GRU_H, "a plane buffer must hold 16 fp32 rows");
    constexpr bool DEDUP = PRESPLIT && REDUCE;           // the aggregation's row plan (LayerArgs::order / tmask) only exists on that path
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int oc = wave * 16 + 4 * grp;
    const int S = a.steps;

    h8v Wi[2][4][3], Wh[2][4][3];                        // [split][k chunk][gate]
    h2_load_weight_tile<1>(a.wih, wave, col, grp, Wi, wsc_ih);
    h2_load_weight_tile<1>(a.whh, wave, col, grp, Wh, csc_hh);
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                asm volatile("" : "+a"(Wh[sp][c][g]));
                if (l8_slot<WL>(sp, c, g) >= 0) Wl[wave][l8_slot<WL>(sp, c, g)][lane] = Wi[sp][c][g];
                else if (c * 3 + g < 8) asm volatile("" : "+a"(Wi[sp][c][g]));     // at two waves per SIMD the file splits 128 / 128: 24 + 8 fragments fill the AGPR half
            }
    __syncthreads();
    for (int i = tid; i < 3 * GRU_H; i += 512) {
        (&csc_hh[0][0])[i] *= (1.f / 16384.f);
        (&bias_s[0][0])[i] = a.bias_gi ? a.bias_gi[i] : 0.f;
    }
    if (tid < GRU_H) csc_hh[3][tid] = a.bhn ? a.bhn[tid] : 0.f;
    if (tid < GRU_H) {
        ln_gb[0][tid] = a.gamma ? a.gamma[tid] : 1.f;
        ln_gb[1][tid] = (a.gamma && a.beta) ? a.beta[tid] : 0.f;
    }
    if (tid < 32) soff_s[tid] = (a.step_off && tid < a.steps) ? a.step_off[tid] : 0;
    __syncthreads();

    const int64_t ntiles = (a.rows + 15) / 16;
    const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};
    const int sr = tid >> 5, sc = (tid & 31) * 4;         // staging role: 16 rows x 32 lanes x one float4
    // PRESPLIT: the aggregation kernel already wrote x as fp16 planes + row scales (the same split, bit for bit): staging is a copy,
    // the ~40 VALU instructions per unit of the max / scale / split leave this matrix-core-bound kernel for an HBM-bound one
    h4v xq1 = {0, 0, 0, 0}, xq2 = {0, 0, 0, 0};
    float xqs = 0.f;
    auto load_x = [&](int64_t tile, int t, f4v &v) {
        if (tile < ntiles) {
            const int64_t row = min(tile * 16 + sr, a.rows - 1);
            if (PRESPLIT) {
                const int64_t rs_ = row * S + t;
                xq1 = *(const h4v *)(a.xp1 + rs_ * GRU_H + sc);
                xq2 = *(const h4v *)(a.xp2 + rs_ * GRU_H + sc);
                xqs = a.xps[rs_];
            } else {
                v = *(const f4v *)(a.x + (a.step_off ? row * a.ld_row + soff_s[t] : (row * S + t) * a.ldx) + sc);
            }
        }
    };
    auto stage_x = [&](int slot, const f4v v) {
        if (PRESPLIT) {
            if ((tid & 31) == 0) xscale[slot][sr] = xqs;
            *(h4v *)(&Xs[slot][0][sr][l8_off(sr, sc)]) = xq1;
            *(h4v *)(&Xs[slot][1][sr][l8_off(sr, sc)]) = xq2;
            return;
        }
        float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
        float scl, inv;
        h2_scale(m, scl, inv);
        if ((tid & 31) == 0) xscale[slot][sr] = scl;
        h4v s0, s1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            _Float16 p, q;
            h2_split<1>(v[j] * inv, p, q); s0[j] = p; s1[j] = q;
        }
        *(h4v *)(&Xs[slot][0][sr][l8_off(sr, sc)]) = s0;
        *(h4v *)(&Xs[slot][1][sr][l8_off(sr, sc)]) = s1;
    };
    // The same staging cut into slices that ride behind the MFMA groups of the running unit (one cross-lane step per slice: its
    // LDS-crossbar latency passes under nine MFMAs instead of standing, five in a row, at the head of every unit where all
    // eight waves wait for it together).  Slice 5 writes the planes of the NEXT unit and requests the x rows of the one after.
    float sx_m = 0.f, sx_p = 0.f;                        // running row maximum, cross-lane value on its way
    auto stage_slice = [&](int i, int slot, f4v &v, bool live, auto request_next) {
        if (PRESPLIT) {
            if (i == 5) {
                if (live) stage_x(slot, v);
                request_next();
            }
            return;
        }
        // row maximum over the 32 lanes of a row: four DPP exchanges inside the 16-lane halves in slice 0 (plain VALU) and ONE cross-half
        // exchange whose LDS-crossbar latency passes under the MFMA groups up to slice 5 (the first version: five ds_bpermute round trips,
        // one per slice).  A maximum does not depend on the order of its operands: same planes, bit for bit.
        if (i == 0) {
            uint32_t u = __float_as_uint(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            u = max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, true));      // quad_perm [1,0,3,2]
            u = max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, true));      // quad_perm [2,3,0,1]
            u = max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x141, 0xf, 0xf, true));     // row_half_mirror
            u = max(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x140, 0xf, 0xf, true));     // row_mirror
            sx_m = __uint_as_float(u);
            sx_p = __shfl_xor(sx_m, 16);
        }
        if (i == 5) {
            sx_m = fmaxf(sx_m, sx_p);
            if (live) {
                float scl, inv;
                h2_scale(sx_m, scl, inv);
                if ((tid & 31) == 0) xscale[slot][sr] = scl;
                h4v s0, s1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 p, q;
                    h2_split<1>(v[j] * inv, p, q); s0[j] = p; s1[j] = q;
                }
                *(h4v *)(&Xs[slot][0][sr][l8_off(sr, sc)]) = s0;
                *(h4v *)(&Xs[slot][1][sr][l8_off(sr, sc)]) = s1;
            }
            request_next();
        }
    };
    // units that need x: all of them, or (DEDUP) the steps whose tmask bit is set — the x pipeline below runs over those only
    auto load_mask = [&](int64_t tile) __attribute__((always_inline)) -> mask_t {
        if constexpr (WIDE) return (mask_t)a.tmask[2 * tile] | ((mask_t)a.tmask[2 * tile + 1] << (WIDE ? 32 : 0));
        else return a.tmask[tile];
    };
    auto tile_mask = [&](int64_t tile) -> mask_t {
        if (DEDUP) return (a.tmask && tile < ntiles) ? load_mask(tile) : MASK_ALL;
        return MASK_ALL;
    };
    mask_t pmask = MASK_ALL;                               // mask of the tile the x pipeline is at
    auto next_unit = [&](int64_t &tile, int &t) {
        do {
            if (++t >= S) { t = 0; tile += nblk; if (DEDUP) pmask = tile_mask(tile); }
        } while (DEDUP && !mask_bit(pmask, t));
    };

    if ((int64_t)bid >= ntiles) return;
    f4v xr;
    int64_t ptile = bid;                           // unit whose x sits in xr
    int pt = 0;
    constexpr bool PLAN_PATH = DEDUP && CTGCN_LAYER_LOOKAHEAD;    // the row-plan path below runs its own x pipeline
    if constexpr (!PLAN_PATH) {
        if (DEDUP) pmask = tile_mask(ptile);
        load_x(ptile, pt, xr);
        stage_x(0, xr);
        next_unit(ptile, pt);
        load_x(ptile, pt, xr);
        __syncthreads();
    }

    int slot = 0;
#ifdef CTGCN_LAYER_TIMELINE
    unsigned long long tl[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // issue of the MFMA stream, gate math (incl. MFMA drain), publish, barrier, tile end, units; row-plan form: see below
    unsigned long long tp = wall_clock64();
#define TL_MARK(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = wall_clock64(); tl[i] += now_ - tp; tp = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define TL_MARK(i)
#endif
    // h planes are double buffered: a unit publishes into Hs[pb] and reads Hs[pb ^ 1]; pb flips after every unit.  The last unit of a
    // tile leaves the summed rows (fp32) in Hs[pb]; their LayerNorm is deferred into the NEXT tile's first unit, which publishes into
    // the other buffer and has no h MFMAs — no barrier pair at the tile end (they were 10 % of the kernel: 237 of 2 260 ns per unit).
    int pb = 0, ln_buf = 0, ln_last = -1;
    int64_t ln_row0 = 0;
    int em_buf = 0, em_last = -1, em_t = 0;                // per-step form: the unit whose rows are still to be emitted
    int64_t em_row0 = 0;
    const bool has_ln_ = a.gamma != nullptr;
    auto ln_row_lds = [&](const float *src, float *dst) {     // = gru_layernorm_row with the weights in LDS (same operations, bit for bit)
        float2 v = *(const float2 *)(src + lane * 2);
        if (has_ln_) {
            const float mean = wave_sum64(v.x + v.y) * (1.0f / GRU_H);
            const float dx = v.x - mean, dy = v.y - mean;
            const float rstd = rsqrtf(wave_sum64(dx * dx + dy * dy) * (1.0f / GRU_H) + a.eps);
            const float2 g = *(const float2 *)(&ln_gb[0][lane * 2]), b = *(const float2 *)(&ln_gb[1][lane * 2]);
            v.x = dx * rstd * g.x + b.x;
            v.y = dy * rstd * g.y + b.y;
        }
        *(float2 *)(dst + (uint32_t)(lane * 2)) = v;
    };
    const int wave_g = __builtin_amdgcn_readfirstlane(wave);      // uniform row numbers: output addresses = scalar base + one 32-bit lane offset
    auto pending_rows = [&]() {
        if (em_last < 0) return;
        for (int r = wave_g * 2; r < wave_g * 2 + 2; ++r)
            if (r <= em_last) ln_row_lds(hrow[REDUCE ? 0 : em_buf][REDUCE ? 0 : r], a.out + ((em_row0 + r) * S + em_t) * GRU_H);
        em_last = -1;
    };
    auto pending_layernorm = [&]() {
        if (ln_last < 0) return;
        for (int r = wave_g * 2; r < wave_g * 2 + 2; ++r)
            if (r <= ln_last) {
                const int64_t orow = (DEDUP && a.order) ? (int64_t)a.order[ln_row0 + r] : ln_row0 + r;
                ln_row_lds((const float *)&Hs[ln_buf][0][0][0] + r * GRU_H, a.out + orow * a.ldo);
            }
        ln_last = -1;
    };
#if CTGCN_LAYER_LOOKAHEAD
    if constexpr (DEDUP) {
        // Inference form (planes + row plan).  The x·W_ih products of a unit do not depend on the recurrence, so they are issued one unit
        // EARLY: at the end of the unit before, after h_t is published and in front of the barrier — where a wave that finishes its gate
        // math first used to wait for the others (414 of 2 260 ns per unit for the older wave of a SIMD, profiles/r02_layer_timeline_summary.txt)
        // it now multiplies.  After the barrier a unit starts with the h·W_hh products alone.  Same MFMAs in the same order per
        // accumulator: bit-identical.  (Also slicing the gate math between the four k chunks of these products — transcendental chains in the
        // MFMAs' shadow — keeps gi, the pre-activations and the new accumulators alive together: 23 scratch reloads inside the loop, 99 instead
        // of 65 ms per window.  Removed.)  The x ring keeps its two slots: slot i % 2 is read here for fresh unit i and the other one, last
        // read one barrier ago, takes the planes of fresh unit i + 1 right behind the MFMAs.
        // Nothing between two barriers may wait on vmcnt except the x staging itself: the counter retires in order, so ANY vector-memory load
        // a wave has to wait for (the next tile's mask, the output row of a finished tile, the LayerNorm weights) also waits for the x planes
        // requested a moment earlier — a full HBM latency, per tile (timeline build: 2.3 us per tile around the LayerNorm, 1.0 us in the x
        // pipeline's tile change, of 14 us).  The scalar cache does not help either (s_load shares lgkmcnt with the LDS reads: the next
        // ds_read's wait exposes its miss).  So a tile's step mask and output rows RIDE WITH ITS FIRST x LOAD (step 0 is always fresh) and
        // are staged with it into a ring of four tiles in LDS (the x pipeline is at most two tiles ahead of the one whose rows leave); the
        // LayerNorm weights sit in LDS.
        // Waves 0-3 (the older wave of each SIMD: it wins the issue arbitration, finishes every phase first and then WAITS at the barrier —
        // 350 of 1 770 ns per unit in the timeline build against 57 for waves 4-7) carry everything that is not on the recurrence's path:
        // the whole x staging (16 rows x 16 lanes x 16 bytes per plane) and the LayerNorm of a finished tile (four rows each).
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const bool stager = wave_u < 4;
        const int qr = (tid >> 4) & 15, qc = (tid & 15) * 8;
        h8v xp1r = {0, 0, 0, 0, 0, 0, 0, 0}, xp2r = {0, 0, 0, 0, 0, 0, 0, 0};
        // row scale, output row, step mask (the last two ride with x_0 only) as ONE four-register tuple, stored with one ds_write_b128: a
        // lone register with a load in flight can end up as the idle half of a packed-math operand pair, and the hardware's wait for
        // that pair is a wait for the whole in-order vmcnt queue — the x planes' HBM latency, at the head of every fresh unit
        // (seen in the ISA of the first version of this pipeline: s_waitcnt vmcnt(0) in front of the gi scaling)
        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
        __shared__ u4v xmeta[2][16];
        u4v xmr = {0, 0, 0xffffffffu, 0};
        int pring = 0;                                    // ring slot of the tile the x pipeline is at
        auto load_xp = [&]() {
            if (ptile < ntiles) {
                const int64_t row = min(ptile * 16 + qr, a.rows - 1);
                const int64_t rs_ = row * S + pt;
                xp1r = *(const h8v *)(a.xp1 + rs_ * GRU_H + qc);
                xp2r = *(const h8v *)(a.xp2 + rs_ * GRU_H + qc);
                xmr[0] = __float_as_uint(a.xps[rs_]);
                if (pt == 0) {
                    xmr[1] = a.order ? (uint32_t)a.order[row] : (uint32_t)row;
                    const mask_t m_ = a.tmask ? load_mask(ptile) : MASK_ALL;
                    xmr[2] = (uint32_t)m_;
                    if constexpr (WIDE) xmr[3] = (uint32_t)((uint64_t)m_ >> 32);
                }
            }
        };
        auto stage_xp = [&](int slot) {                   // the unit in the registers -> planes of `slot`; a tile's step 0 brings its plan entries
            if (ptile < ntiles) {
                if ((tid & 15) == 0) xmeta[slot][qr] = xmr;
                *(h8v *)(&Xs[slot][0][qr][l8_off(qr, qc)]) = xp1r;
                *(h8v *)(&Xs[slot][1][qr][l8_off(qr, qc)]) = xp2r;
                if (pt == 0) {
                    if ((tid & 15) == 0) ord_s[pring][qr] = (int32_t)xmr[1];
                    const mask_t m_ = WIDE ? (mask_t)(((uint64_t)xmr[3] << 32) | xmr[2]) : (mask_t)xmr[2];
                    if (tid == 0) msk_s[pring] = m_;
                    pmask = uniform_mask(m_);
                }
            }
            do {
                if (++pt >= S) { pt = 0; ptile += nblk; pring = (pring + 1) & 3; break; }     // step 0 of a tile is always fresh
            } while (!mask_bit(pmask, pt));
        };
        // ---- GLDS build (A/B): the planes of a fresh unit go global -> LDS by DMA (global_load_lds_dwordx4: lane L of stager wave w lands at
        // byte 16 L of rows 4w .. 4w + 3 of the slot, so it REQUESTS segment (L & 15) ^ row of its row: the XOR swizzle moves to the source
        // address), the row scales and output rows by global_load_lds_dword; no staging registers, no ds_write pass.  hipcc drains the DMA
        // queue (vmcnt(0)) in front of every s_barrier while such a load is in flight, so a unit is requested right AFTER a barrier — at the
        // head of a compute unit — and has that unit's time to land; ring of three slots (the slot x_products read last is free after the
        // barrier), at most two units ahead, one request per compute unit.  A tile's step mask is an ordinary load requested with its step 0
        // and looked at one tick later (mask_pending): where the pipeline goes after step 0 is only known then.
        __shared__ float xsc_s[XNS][16];
        int n_issued = 0, n_consumed = 0;
        bool mask_pending = false;
        mask_t mreg = MASK_ALL;
        int mring = 0;
        auto advance_p = [&]() {
            do {
                if (++pt >= S) { pt = 0; ptile += nblk; pring = (pring + 1) & 3; break; }     // step 0 of a tile is always fresh
            } while (!mask_bit(pmask, pt));
        };
        auto tick = [&]() {                               // stager waves, once per compute unit (and twice in the prologue)
            if constexpr (GLDS) {
                if (mask_pending) {                       // the mask requested one tick ago (a barrier with vmcnt(0) lies in between)
                    pmask = uniform_mask(mreg);
                    if (tid == 0) msk_s[mring] = pmask;
                    mask_pending = false;
                    advance_p();
                }
                if (n_issued - n_consumed >= 2 || ptile >= ntiles) return;
                const int islot = n_issued % 3;
                const int r = wave_u * 4 + (lane >> 4);
                const int64_t row = min(ptile * 16 + r, a.rows - 1);
                const int64_t rs_ = row * S + pt;
                const int q = ((lane & 15) ^ r) & 15;
                if (pt == 0 && a.tmask) { mreg = load_mask(ptile); }        // older than the DMA requests below: its wait does not wait for them
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.xp1 + rs_ * GRU_H + q * 8),
                                                 (__attribute__((address_space(3))) void *)(&Xs[islot][0][wave_u * 4][0]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.xp2 + rs_ * GRU_H + q * 8),
                                                 (__attribute__((address_space(3))) void *)(&Xs[islot][1][wave_u * 4][0]), 16, 0, 0);
                if (lane < 4) {
                    const int64_t row4 = min(ptile * 16 + wave_u * 4 + lane, a.rows - 1);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.xps + row4 * S + pt),
                                                     (__attribute__((address_space(3))) void *)(&xsc_s[islot][wave_u * 4]), 4, 0, 0);
                    if (pt == 0) {
                        if (a.order)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.order + row4),
                                                             (__attribute__((address_space(3))) void *)(&ord_s[pring][wave_u * 4]), 4, 0, 0);
                        else ord_s[pring][wave_u * 4 + lane] = (int32_t)row4;
                    }
                }
                ++n_issued;
                if (pt == 0) {
                    if (a.tmask) { mask_pending = true; mring = pring; }
                    else { pmask = MASK_ALL; if (tid == 0) msk_s[pring] = MASK_ALL; advance_p(); }
                } else advance_p();
            }
        };
        if constexpr (GLDS) {
            if (stager) { tick(); tick(); }
        } else if (stager) {
            load_xp();
            stage_xp(0);
            load_xp();
        }
        __syncthreads();
        f4v acc0[3] = {zero4, zero4, zero4};
        float rs_n = 0.f;
        int xslot = 0;                                    // slot of the next unit that brings a new x
        h8v xo1, xo2;                                     // x operand fragments of the running chunk
        auto x_begin = [&]() {
#pragma unroll
            for (int g = 0; g < 3; ++g) acc0[g] = zero4;
            if constexpr (GLDS) rs_n = xsc_s[xslot][col];
            else rs_n = __uint_as_float(xmeta[xslot][col][0]);
            xo1 = *(const h8v *)(&Xs[xslot][0][col][l8_off(col, 8 * grp)]);
            xo2 = *(const h8v *)(&Xs[xslot][1][col][l8_off(col, 8 * grp)]);
        };
        auto x_chunk = [&](auto ctag) {
            constexpr int c = decltype(ctag)::value;
            h8v wr[3];
#pragma unroll
            for (int g = 0; g < 3; ++g) wr[g] = l8_slot<WL>(1, c, g) >= 0 ? Wl[wave][l8_slot<WL>(1, c, g) < 0 ? 0 : l8_slot<WL>(1, c, g)][lane] : Wi[1][c][g];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const h8v w1 = l8_slot<WL>(0, c, g) >= 0 ? Wl[wave][l8_slot<WL>(0, c, g) < 0 ? 0 : l8_slot<WL>(0, c, g)][lane] : Wi[0][c][g];
                acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xo2, acc0[g], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[g], xo1, acc0[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const h8v w1 = l8_slot<WL>(0, c, g) >= 0 ? Wl[wave][l8_slot<WL>(0, c, g) < 0 ? 0 : l8_slot<WL>(0, c, g)][lane] : Wi[0][c][g];
                acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xo1, acc0[g], 0, 0, 0);
            }
            if (c < 3) {
                __builtin_amdgcn_sched_barrier(0);
                xo1 = *(const h8v *)(&Xs[xslot][0][col][l8_off(col, (c + 1) * 32 + 8 * grp)]);
                xo2 = *(const h8v *)(&Xs[xslot][1][col][l8_off(col, (c + 1) * 32 + 8 * grp)]);
            }
        };
        auto x_end = [&]() {
            // the planes of the fresh unit after this one (in registers since the last call) go to the other slot; request the one after
            if constexpr (GLDS) {
                xslot = xslot == 2 ? 0 : xslot + 1;
                ++n_consumed;
            } else {
                if (stager) {
                    stage_xp(xslot ^ 1);
                    load_xp();
                }
                xslot ^= 1;
            }
        };
        auto x_products = [&]() {
            x_begin();
            {
            x_chunk(std::integral_constant<int, 0>{}); x_chunk(std::integral_constant<int, 1>{});
            x_chunk(std::integral_constant<int, 2>{}); x_chunk(std::integral_constant<int, 3>{});
            }
            TL_MARK(3)
            x_end();
        };
        x_products();                                     // the block's first unit
        __syncthreads();                                  // its x_end staged the SECOND fresh unit, which the first unit's x_products reads — every other
                                                          // staging is a barrier ahead of its reader by construction, this one is not (found by
                                                          // tools/stress_group.py: one forward in 1 200 differed in four rows once waves 0-3 alone staged)
        int pb = 0, ln_buf = 0, ln_last = -1;
        int cring = 0, ln_ring = 0;                       // ring slots of the running tile and of the one whose rows are about to leave
        const bool has_ln = a.gamma != nullptr;
        auto pending_layernorm = [&]() {
            if (ln_last < 0) return;
            if (!stager) { ln_last = -1; return; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave_u * 4 + i;
                if (r <= ln_last) {
                    float2 v = *(const float2 *)((const float *)&Hs[ln_buf][0][0][0] + r * GRU_H + lane * 2);
                    if (has_ln) {                          // = gru_layernorm_vals with the weights in LDS
                        const float mean = wave_sum64(v.x + v.y) * (1.0f / GRU_H);
                        const float dx = v.x - mean, dy = v.y - mean;
                        const float rstd = rsqrtf(wave_sum64(dx * dx + dy * dy) * (1.0f / GRU_H) + a.eps);
                        const float2 g = *(const float2 *)(&ln_gb[0][lane * 2]), b = *(const float2 *)(&ln_gb[1][lane * 2]);
                        v.x = dx * rstd * g.x + b.x;
                        v.y = dy * rstd * g.y + b.y;
                    }
                    *(float2 *)(a.out + (int64_t)__builtin_amdgcn_readfirstlane(ord_s[ln_ring][r]) * a.ldo + lane * 2) = v;
                }
            }
            ln_last = -1;
        };
        for (int64_t tile = bid; tile < ntiles; tile += nblk, cring = (cring + 1) & 3) {
            const int64_t row0 = tile * 16;
            const int last = (int)min((int64_t)16, a.rows - row0) - 1;
            f4v hprev = zero4, hsum = zero4;
            const mask_t tmask = uniform_mask(msk_s[cring]);
            f4v gi[3] = {zero4, zero4, zero4};
            for (int t = 0; t < S; ++t) {
                if constexpr (GLDS) { if (stager) tick(); }     // the next fresh unit's DMA, right behind the barrier that ended the last unit
                if (t == 0) pending_layernorm();          // the previous tile's rows (its last unit ended with a barrier)
                TL_MARK(5)                                // row-plan form: [0] h products issued, [1] gate math, [2] publish, [3] x products issued, [4] barrier, [5] LayerNorm, [6] units, [7] fresh, [8] x staging + next request
                if (mask_bit(tmask, t)) {
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        gi[g] = acc0[g] * (*(const f4v *)(&wsc_ih[g][oc]) * rs_n) + *(const f4v *)(&bias_s[g][oc]);
                }
                f4v ach[3] = {zero4, zero4, zero4};
                if (t > 0) {
                    const int hp = pb ^ 1;                // the buffer step t-1 published into
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const h8v h1 = *(const h8v *)(&Hs[hp][0][col][l8_off(col, c * 32 + 8 * grp)]);
                        const h8v h2 = *(const h8v *)(&Hs[hp][1][col][l8_off(col, c * 32 + 8 * grp)]);
                        CTGCN_H2_MFMA1(Wh, c, h1, h2, ach)
                    }
                }
                const f4v csc[3] = {*(const f4v *)(&csc_hh[0][oc]), *(const f4v *)(&csc_hh[1][oc]), *(const f4v *)(&csc_hh[2][oc])};
                const f4v b_hn = *(const f4v *)(&csc_hh[3][oc]);
                // the next unit's x·W_ih, if it brings a new x (a repeat keeps gi): this tile's next step, or step 0 of the block's next tile
                const bool next_fresh = t + 1 < S ? mask_bit(tmask, t + 1) : tile + nblk < ntiles;
                TL_MARK(0)
                f4v h, rv4, zv4, nv4, an4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rv = gru_sigmoid(fmaf(ach[0][j], csc[0][j], gi[0][j]));
                    const float zv = gru_sigmoid(fmaf(ach[1][j], csc[1][j], gi[1][j]));
                    const float an = fmaf(ach[2][j], csc[2][j], b_hn[j]);
                    const float nv = gru_tanh(fmaf(rv, an, gi[2][j]));
                    h[j] = nv + zv * (hprev[j] - nv);
                    if (SAVE) { rv4[j] = rv; zv4[j] = zv; nv4[j] = nv; an4[j] = an; }
                }
                if (SAVE && col <= last) {                // training's recompute pass: gates and raw h of this step, position order
                    // uniform 64-bit base (tile, step) + ONE 32-bit lane offset per stream: with per-lane 64-bit addresses the recompute pass
                    // spilled its loop-invariant address pairs and reloaded them behind the x requests (scratch is vector memory: in-order vmcnt)
                    const int ng = a.gates3 ? 3 : 4;
                    float *gb = a.gates + ((int64_t)row0 * S + t) * (ng * GRU_H);
                    float *hb = a.hseq + ((int64_t)row0 * S + t) * GRU_H;
                    const uint32_t go = (uint32_t)(col * S) * (uint32_t)(ng * GRU_H) + (uint32_t)oc;
                    const uint32_t ho = (uint32_t)(col * S) * (uint32_t)GRU_H + (uint32_t)oc;
                    if (a.gates3) {
                        *(f4v *)(gb + go) = rv4; *(f4v *)(gb + go + GRU_H) = zv4; *(f4v *)(gb + go + 2 * GRU_H) = an4;
                    } else {
                        *(f4v *)(gb + go) = rv4; *(f4v *)(gb + go + GRU_H) = zv4; *(f4v *)(gb + go + 2 * GRU_H) = nv4; *(f4v *)(gb + go + 3 * GRU_H) = an4;
                    }
                    *(f4v *)(hb + ho) = h;
                }
                hprev = h;
                hsum = t > 0 ? hsum + h : h;
#ifdef CTGCN_LAYER_TIMELINE
                asm volatile("" :: "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
#endif
                TL_MARK(1)
                if (t + 1 < S) {                          // fp16x2 planes of h·2^14 for the next step (the buffer nobody reads in this unit)
                    h4v p, q;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        _Float16 x, y;
                        h2_split<1>(h[j] * 16384.f, x, y);
                        p[j] = x; q[j] = y;
                    }
                    *(h4v *)(&Hs[pb][0][col][l8_off(col, oc)]) = p;
                    *(h4v *)(&Hs[pb][1][col][l8_off(col, oc)]) = q;
                } else if (SAVE) {                        // recompute pass: the pre-LayerNorm sum leaves as it is (its backward needs it)
                    if (col <= last) *(f4v *)((a.presum + (int64_t)row0 * GRU_H) + (uint32_t)(col * GRU_H + oc)) = hsum;
                } else {                                  // last step: that buffer takes the summed rows (fp32) for the LayerNorm instead
                    *(f4v *)((float *)&Hs[pb][0][0][0] + col * GRU_H + oc) = hsum;
                    ln_buf = pb; ln_last = last; ln_ring = cring;
                }
                __builtin_amdgcn_sched_barrier(0);
                TL_MARK(2)
                if (next_fresh) x_products();
                TL_MARK(8)
                __syncthreads();
                TL_MARK(4)
#ifdef CTGCN_LAYER_TIMELINE
                ++tl[6];
                tl[7] += next_fresh ? 1 : 0;
#endif
                pb ^= 1;
            }
        }
        pending_layernorm();
#ifdef CTGCN_LAYER_TIMELINE
        if (a.timeline && lane == 0)
            for (int i = 0; i < 12; ++i) a.timeline[((size_t)bid * 8 + wave) * 12 + i] = tl[i];
#endif
        return;
    }
#endif
    for (int64_t tile = bid; tile < ntiles; tile += nblk) {
        const int64_t row0 = tile * 16;
        const int last = (int)min((int64_t)16, a.rows - row0) - 1;
        f4v hprev = zero4, hsum = zero4;
        const mask_t tmask = tile_mask(tile);
        f4v gi[3] = {zero4, zero4, zero4};               // x_t·W_ih + b of the last step that brought a new x (kept over its repeats)
        for (int t = 0; t < S; ++t) {
            const bool fresh = !DEDUP || mask_bit(tmask, t);
            if (REDUCE && t == 0) pending_layernorm();
            if (!REDUCE) pending_rows();                   // the previous unit's rows (staged before its barrier)
            // ---- x of the next unit (registers -> planes of the other slot, then the request for the unit after it) is staged in
            // slices behind this unit's MFMA groups: see stage_slice
            const bool stage_live = ptile < ntiles;
            auto request_next = [&]() {
                next_unit(ptile, pt);
                load_x(ptile, pt, xr);
            };
            // ---- this unit
            f4v acc0[3] = {zero4, zero4, zero4}, ach[3] = {zero4, zero4, zero4};
            auto body = [&](auto with_h_tag) {
                constexpr bool with_h = decltype(with_h_tag)::value;
                const int hp = pb ^ 1;                    // the buffer step t-1 published into
                // issue order pinned as in gru_layer_h2_kernel: per k chunk the h planes and the residual-plane fragments are
                // requested before the x MFMAs, the next chunk's x planes before the h MFMAs
                h8v x1 = *(const h8v *)(&Xs[slot][0][col][l8_off(col, 8 * grp)]);
                h8v x2 = *(const h8v *)(&Xs[slot][1][col][l8_off(col, 8 * grp)]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    h8v h1, h2, wr[3];
                    if (with_h) {
                        h1 = *(const h8v *)(&Hs[hp][0][col][l8_off(col, c * 32 + 8 * grp)]);
                        h2 = *(const h8v *)(&Hs[hp][1][col][l8_off(col, c * 32 + 8 * grp)]);
                    }
#pragma unroll
                    for (int g = 0; g < 3; ++g) wr[g] = l8_slot<WL>(1, c, g) >= 0 ? Wl[wave][l8_slot<WL>(1, c, g) < 0 ? 0 : l8_slot<WL>(1, c, g)][lane] : Wi[1][c][g];
                    __builtin_amdgcn_sched_barrier(0);
                    // = CTGCN_H2_MFMA1(Wi, c, x1, x2, acc0): one accumulator, small terms first (w1·x2, w2·x1, then w1·x1); LDS-resident
                    // fragments of the leading plane are read on the spot
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const h8v w1 = l8_slot<WL>(0, c, g) >= 0 ? Wl[wave][l8_slot<WL>(0, c, g) < 0 ? 0 : l8_slot<WL>(0, c, g)][lane] : Wi[0][c][g];
                        acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x2, acc0[g], 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[g], x1, acc0[g], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const h8v w1 = l8_slot<WL>(0, c, g) >= 0 ? Wl[wave][l8_slot<WL>(0, c, g) < 0 ? 0 : l8_slot<WL>(0, c, g)][lane] : Wi[0][c][g];
                        acc0[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, x1, acc0[g], 0, 0, 0);
                    }
                    if (c < 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        x1 = *(const h8v *)(&Xs[slot][0][col][l8_off(col, (c + 1) * 32 + 8 * grp)]);
                        x2 = *(const h8v *)(&Xs[slot][1][col][l8_off(col, (c + 1) * 32 + 8 * grp)]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (2 * c < 6) stage_slice(2 * c, slot ^ 1, xr, stage_live, request_next);
                    __builtin_amdgcn_sched_barrier(0);
                    if (with_h) { CTGCN_H2_MFMA1(Wh, c, h1, h2, ach) }
                    __builtin_amdgcn_sched_barrier(0);
                    if (2 * c + 1 < 6) stage_slice(2 * c + 1, slot ^ 1, xr, stage_live, request_next);
                }
            };
            if (fresh) {
                const float rs = xscale[slot][col];
                if (t > 0) body(std::true_type{}); else body(std::false_type{});
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    gi[g] = acc0[g] * (*(const f4v *)(&wsc_ih[g][oc]) * rs) + *(const f4v *)(&bias_s[g][oc]);
            } else {
                // x_t = x_{t-1} for the whole tile: gi stands, only h_{t-1}·W_hh is new (t > 0: bit 0 of a mask is always set)
                const int hp = pb ^ 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const h8v h1 = *(const h8v *)(&Hs[hp][0][col][l8_off(col, c * 32 + 8 * grp)]);
                    const h8v h2 = *(const h8v *)(&Hs[hp][1][col][l8_off(col, c * 32 + 8 * grp)]);
                    CTGCN_H2_MFMA1(Wh, c, h1, h2, ach)
                }
            }
            TL_MARK(0)
            const f4v csc[3] = {*(const f4v *)(&csc_hh[0][oc]), *(const f4v *)(&csc_hh[1][oc]), *(const f4v *)(&csc_hh[2][oc])};
            const f4v b_hn = *(const f4v *)(&csc_hh[3][oc]);
            f4v h, rv4, zv4, nv4, an4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float rv = gru_sigmoid(fmaf(ach[0][j], csc[0][j], gi[0][j]));
                const float zv = gru_sigmoid(fmaf(ach[1][j], csc[1][j], gi[1][j]));
                const float an = fmaf(ach[2][j], csc[2][j], b_hn[j]);
                const float nv = gru_tanh(fmaf(rv, an, gi[2][j]));
                h[j] = nv + zv * (hprev[j] - nv);
                if (SAVE) { rv4[j] = rv; zv4[j] = zv; nv4[j] = nv; an4[j] = an; }
            }
            if (SAVE && col <= last) {                    // the gates of this step for the backward recurrence (training: recompute pass)
                float *gb = a.gates + ((int64_t)row0 * S + t) * (4 * GRU_H);       // scalar base + one 32-bit lane offset (see the row-plan path)
                const uint32_t go = (uint32_t)(col * S) * (uint32_t)(4 * GRU_H) + (uint32_t)oc;
                *(f4v *)(gb + go) = rv4; *(f4v *)(gb + go + GRU_H) = zv4; *(f4v *)(gb + go + 2 * GRU_H) = nv4; *(f4v *)(gb + go + 3 * GRU_H) = an4;
            }
            hprev = h;
            if (REDUCE) hsum = t > 0 ? hsum + h : h;
            else {
                *(f4v *)(&hrow[REDUCE ? 0 : pb][REDUCE ? 0 : col][oc]) = h;
                em_buf = pb; em_last = last; em_t = t; em_row0 = row0;
            }
#ifdef CTGCN_LAYER_TIMELINE
            asm volatile("" :: "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));
#endif
            TL_MARK(1)
            if (t + 1 < S) {                              // fp16x2 planes of h·2^14 for the next step (the buffer nobody reads in this unit)
                h4v p, q;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 x, y;
                    h2_split<1>(h[j] * 16384.f, x, y);
                    p[j] = x; q[j] = y;
                }
                *(h4v *)(&Hs[pb][0][col][l8_off(col, oc)]) = p;
                *(h4v *)(&Hs[pb][1][col][l8_off(col, oc)]) = q;
            } else if (REDUCE) {                          // last step: that buffer takes the summed rows (fp32) for the LayerNorm instead
                *(f4v *)((float *)&Hs[pb][0][0][0] + col * GRU_H + oc) = hsum;
                ln_buf = pb; ln_last = last; ln_row0 = row0;
            }
            TL_MARK(2)
            __syncthreads();
            TL_MARK(3)
#ifdef CTGCN_LAYER_TIMELINE
            ++tl[5];
#endif
            if (fresh) slot ^= 1;
            pb ^= 1;
        }
        TL_MARK(4)
    }
    if (REDUCE) pending_layernorm();       // the block's last tile (its rows are visible: the last unit ended with a barrier)
    else pending_rows();
#ifdef CTGCN_LAYER_TIMELINE
    if (a.timeline && lane == 0)
        for (int i = 0; i < 12; ++i) a.timeline[((size_t)bid * 8 + wave) * 12 + i] = tl[i];
#endif
#undef TL_MARK
}

template <bool PRESPLIT, bool REDUCE, bool SAVE = false, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void gru_layer8_h2_kernel(const LayerArgs a)
{
    gru_layer8_h2_body<PRESPLIT, REDUCE, SAVE, WIDE>(a, (int)blockIdx.x, (int)gridDim.x);
}
// One launch for the width-128 CoreDiffusion layer of EVERY snapshot of a window (small graphs: a snapshot alone is 25-500 us of kernel, most
// of it ramp and tail).  Snapshots own their weights (reference models.py:225-231: duffision_list[t]), so a block serves one snapshot:
// blockmap[b] = {snapshot, block index among that snapshot's blocks, their number}; table[snapshot] = that snapshot's arguments.
__global__ __launch_bounds__(512, 2) void gru_layer8_h2_group_kernel(const LayerArgs *__restrict__ table, const int32_t *__restrict__ blockmap)
{
    const int g = __builtin_amdgcn_readfirstlane(blockmap[3 * blockIdx.x]);
    const int bid = __builtin_amdgcn_readfirstlane(blockmap[3 * blockIdx.x + 1]);
    const int nblk = __builtin_amdgcn_readfirstlane(blockmap[3 * blockIdx.x + 2]);
    const LayerArgs a = table[g];
    gru_layer8_h2_body<true, true, false>(a, bid, nblk);
}

// ------------------------------------------------------------------------------------------------
// LSTM recurrence (rnn_type = 'LSTM', reference layers.py:27-28 / models.py:234-235), same fusion as the GRU:
//   gates = GI_t + h_{t-1}·W_hhᵀ (order i,f,g,o; both biases are already in GI);  c_t = σ(f)·c_{t-1} + σ(i)·tanh(g);
//   h_t = σ(o)·tanh(c_t);   out = LayerNorm(Σ_t h_t)  or  LayerNorm(h_t) per step.
// Exact fp32 (v_mfma_f32_16x16x4_f32).  Wave w owns hidden units [16w,16w+16) of all four gates: 128 VGPRs of B
// operands; 32 rows per block iteration; the cell state never leaves registers.
// ------------------------------------------------------------------------------------------------
constexpr int LSTM_BM = 32;
constexpr int LSTM_RT = LSTM_BM / 16;

template <bool REDUCE, bool SAVE = false>
__global__ __launch_bounds__(512, 2) void lstm_seq_kernel(const GruArgs a)
{
    __shared__ float hbuf[2][LSTM_BM][GRU_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int hid = wave * 16 + col;
    const int steps = a.steps;

    float W[4][32];          // W[g][kk] = W_hh[g*128 + hid][32*grp + kk]
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f4v *src = (const f4v *)(a.whh + (int64_t)(g * GRU_H + hid) * GRU_H + 32 * grp);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f4v v = src[q];
            W[g][4 * q + 0] = v.x; W[g][4 * q + 1] = v.y; W[g][4 * q + 2] = v.z; W[g][4 * q + 3] = v.w;
        }
    }
    const int64_t ntiles = (a.rows + LSTM_BM - 1) / LSTM_BM;
    const int gstride = steps * 4 * GRU_H;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * LSTM_BM;
        const float *gi_tile = a.gi + row0 * gstride + hid;
        const int last = (int)min((int64_t)LSTM_BM, a.rows - row0) - 1;
        float creg[LSTM_RT][4], hsum[LSTM_RT][4];
#pragma unroll
        for (int rt = 0; rt < LSTM_RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) { creg[rt][i] = 0.f; hsum[rt][i] = 0.f; }

        for (int t = 0; t < steps; ++t) {
            const float(*hprev)[GRU_PITCH] = hbuf[(t - 1) & 1];
            float(*hcur)[GRU_PITCH] = hbuf[t & 1];
#pragma unroll
            for (int rt = 0; rt < LSTM_RT; ++rt) {
                float gi[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float *p = gi_tile + min(rt * 16 + grp * 4 + i, last) * gstride + t * 4 * GRU_H;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gi[g][i] = p[g * GRU_H];
                }
                f4v acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
                if (t > 0) {
                    float av[32];
                    const f4v *src = (const f4v *)(&hprev[rt * 16 + col][32 * grp]);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const f4v v = src[q];
                        av[4 * q + 0] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
                    }
#pragma unroll
                    for (int kk = 0; kk < 32; ++kk)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], W[g][kk], acc[g], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float ig = gru_sigmoid(gi[0][i] + acc[0][i]);
                    const float fg = gru_sigmoid(gi[1][i] + acc[1][i]);
                    const float gg = gru_tanh(gi[2][i] + acc[2][i]);
                    const float og = gru_sigmoid(gi[3][i] + acc[3][i]);
                    const float c = fg * creg[rt][i] + ig * gg;
                    const float h = og * gru_tanh(c);
                    creg[rt][i] = c;
                    hsum[rt][i] += h;
                    hcur[rt * 16 + grp * 4 + i][hid] = h;
                    if (SAVE && rt * 16 + grp * 4 + i <= last) {       // training (recompute pass): i, f, g, o, c of this step for lstm_seq_bwd_kernel
                        float *gp = a.gates + ((row0 + rt * 16 + grp * 4 + i) * steps + t) * (5 * GRU_H) + hid;
                        gp[0] = ig; gp[GRU_H] = fg; gp[2 * GRU_H] = gg; gp[3 * GRU_H] = og; gp[4 * GRU_H] = c;
                    }
                }
            }
            __syncthreads();
            if (!REDUCE)
                for (int r = wave; r <= last; r += 8)
                    gru_layernorm_row(hcur[r], a.out + ((row0 + r) * steps + t) * GRU_H, lane, a.gamma, a.beta, a.eps);
        }
        if (REDUCE) {
            float(*sbuf)[GRU_PITCH] = hbuf[steps & 1];
#pragma unroll
            for (int rt = 0; rt < LSTM_RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) sbuf[rt * 16 + grp * 4 + i][hid] = hsum[rt][i];
            __syncthreads();
            for (int r = wave; r <= last; r += 8)
                gru_layernorm_row(sbuf[r], a.out + (row0 + r) * a.ldo, lane, a.gamma, a.beta, a.eps);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the LSTM recurrence (what autograd derives for nn.LSTM, rnn_type = 'LSTM': layers.py:27-28 / models.py:234-235).
// Walks t = steps-1 .. 0 with dh = dh_seq[t] (or the broadcast dh_sum) + the recurrent term, dc carried in registers:
//   tc = tanh(c_t); do = dh tc; dc += dh o (1 - tc²); di = dc g; df = dc c_{t-1}; dg = dc i; dc_{t-1} = dc f
//   dGI[t] = (di i(1-i), df f(1-f), dg (1-g²), do o(1-o))      dh_{t-1} = dGI[t]·W_hh
// Same shape as gru_seq_bwd_kernel, exact fp32 (v_mfma_f32_16x16x4_f32): wave w owns hidden units [16w,16w+16) of dh and keeps
// W_hh[:, 16w..] (512 x 16) in 128 VGPRs as MFMA B operands; the block's dGI rows go through LDS (double buffered) as A operands.
// d x and the weight gradients are plain GEMMs over the materialised dGI and stay with the caller's BLAS.
// ------------------------------------------------------------------------------------------------
constexpr int LSTMB_BM = 32;
constexpr int LSTMB_RT = LSTMB_BM / 16;
constexpr int LSTMB_PITCH = 4 * GRU_H + 4;

struct LstmBwdArgs {
    int64_t rows;
    int32_t steps;
    const float *gates;    // [rows, steps, 5, 128]: i, f, g, o, c
    const float *dh_seq;   // [rows, steps, 128] or null
    const float *dh_sum;   // [rows, 128] added at every step, or null
    const float *whh;      // [512, 128]
    float *dgi;            // [rows, steps, 512]
    float *bias_partial;   // [gridDim.x, 512] per-block column sums of d_gi, or null
};

__global__ __launch_bounds__(512, 2) void lstm_seq_bwd_kernel(const LstmBwdArgs a)
{
    __shared__ float gbuf[2][LSTMB_BM][LSTMB_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int hid = wave * 16 + col;
    const int steps = a.steps;

    // B operands: dh_prev[:, hid] = sum_k dGI[:, k] W_hh[k][hid];  k = g*128 + 32*grp + kk
    float W[4][32];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) W[g][kk] = a.whh[(int64_t)(g * GRU_H + 32 * grp + kk) * GRU_H + hid];

    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t ntiles = (a.rows + LSTMB_BM - 1) / LSTMB_BM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * LSTMB_BM;
        const int last = (int)min((int64_t)LSTMB_BM, a.rows - row0) - 1;
        float drec[LSTMB_RT][4], dcar[LSTMB_RT][4];
#pragma unroll
        for (int rt = 0; rt < LSTMB_RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) { drec[rt][i] = 0.f; dcar[rt][i] = 0.f; }

        for (int t = steps - 1; t >= 0; --t) {
            float(*gcur)[LSTMB_PITCH] = gbuf[t & 1];
#pragma unroll
            for (int rt = 0; rt < LSTMB_RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r_ = rt * 16 + grp * 4 + i;
                    const int64_t row = row0 + min(r_, last);
                    const float *gp = a.gates + (row * steps + t) * (5 * GRU_H) + hid;
                    const float ig = gp[0], fg = gp[GRU_H], gg = gp[2 * GRU_H], og = gp[3 * GRU_H], c = gp[4 * GRU_H];
                    const float cprev = t > 0 ? gp[4 * GRU_H - 5 * GRU_H] : 0.f;       // c of step t-1: one step back in the same row
                    float dh = drec[rt][i];
                    if (a.dh_seq) dh += a.dh_seq[(row * steps + t) * GRU_H + hid];
                    if (a.dh_sum) dh += a.dh_sum[row * GRU_H + hid];
                    const float tc = gru_tanh(c);
                    const float dc = dcar[rt][i] + dh * og * (1.f - tc * tc);
                    const float dai = dc * gg * ig * (1.f - ig);
                    const float daf = dc * cprev * fg * (1.f - fg);
                    const float dag = dc * ig * (1.f - gg * gg);
                    const float dao = dh * tc * og * (1.f - og);
                    dcar[rt][i] = dc * fg;
                    drec[rt][i] = 0.f;                           // no direct h path in an LSTM: the W_hh term is added after the MFMAs
                    gcur[r_][hid] = dai;
                    gcur[r_][GRU_H + hid] = daf;
                    gcur[r_][2 * GRU_H + hid] = dag;
                    gcur[r_][3 * GRU_H + hid] = dao;
                    if (r_ <= last) {
                        float *o = a.dgi + (row * steps + t) * (4 * GRU_H) + hid;
                        o[0] = dai; o[GRU_H] = daf; o[2 * GRU_H] = dag; o[3 * GRU_H] = dao;
                        bsum[0] += dai; bsum[1] += daf; bsum[2] += dag; bsum[3] += dao;
                    }
                }
            if (t == 0) break;                                   // h_{-1} is the constant 0: nothing to propagate
            __syncthreads();
#pragma unroll
            for (int rt = 0; rt < LSTMB_RT; ++rt) {
                f4v acc0 = f4v{0.f, 0.f, 0.f, 0.f}, acc1 = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float av[32];
                    const f4v *src = (const f4v *)(&gcur[rt * 16 + col][g * GRU_H + 32 * grp]);
#pragma unroll
                    for (int qd = 0; qd < 8; ++qd) {
                        const f4v v = src[qd];
                        av[4 * qd + 0] = v.x; av[4 * qd + 1] = v.y; av[4 * qd + 2] = v.z; av[4 * qd + 3] = v.w;
                    }
#pragma unroll
                    for (int kk = 0; kk < 32; kk += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], W[g][kk], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk + 1], W[g][kk + 1], acc1, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) drec[rt][i] = acc0[i] + acc1[i];
            }
        }
        __syncthreads();       // LDS is reused by the next tile
    }
    if (a.bias_partial) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bsum[i] += __shfl_xor(bsum[i], 16);
            bsum[i] += __shfl_xor(bsum[i], 32);
        }
        if (grp == 0) {
            float *o = a.bias_partial + (int64_t)blockIdx.x * (4 * GRU_H) + hid;
            o[0] = bsum[0]; o[GRU_H] = bsum[1]; o[2 * GRU_H] = bsum[2]; o[3 * GRU_H] = bsum[3];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the GRU recurrence (what autograd derives for nn.GRU at layers.py:59 / models.py:249).
// Walks t = steps-1 .. 0 with dh = dh_seq[t] (or the broadcast dh_sum) + the recurrent term carried in registers:
//   dn = dh(1-z); dz = dh(h_{t-1}-n); da_n = dn(1-n²); da_z = dz z(1-z); da_r = da_n q r(1-r)
//   dGI[t] = (da_r, da_z, da_n)      dGHn[t] = da_n r      dh_{t-1} = dh z + (da_r, da_z, da_n r)·W_hh
// Same block shape as the forward: wave w owns hidden units [16w,16w+16) of dh and keeps W_hh[:, 16w..] (384 x 16)
// in 96 VGPRs as MFMA B operands; the block's dGH rows go through LDS (double buffered) as A operands.
// The weight gradients are plain GEMMs over the materialised dGI / dGHn and stay with the caller's BLAS.
// ------------------------------------------------------------------------------------------------
constexpr int GRUB_BM = 32;
constexpr int GRUB_RT = GRUB_BM / 16;
constexpr int GRUB_PITCH = 3 * GRU_H + 4;

struct GruBwdArgs {
    int64_t rows;
    int32_t steps;
    const float *gates;    // [rows, steps, 4, 128]
    const float *hseq;     // [rows, steps, 128]
    const float *dh_seq;   // [rows, steps, 128] or null
    const float *dh_sum;   // [rows, 128] added at every step, or null
    const float *whh;      // [384, 128]
    float *dgi;            // [rows, steps, 384]
    float *dghn;           // [rows, steps, 128]
    float *bias_partial;   // [gridDim.x, 512] per-block column sums of (d_gi | d_ghn), or null
};

__global__ __launch_bounds__(512, 2) void gru_seq_bwd_kernel(const GruBwdArgs a)
{
    __shared__ float gbuf[2][GRUB_BM][GRUB_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int hid = wave * 16 + col;
    const int steps = a.steps;

    // B operands: dh_prev[:, hid] = sum_k dGH[:, k] W_hh[k][hid];  k = g*128 + 32*grp + kk
    float W[3][32];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) W[g][kk] = a.whh[(int64_t)(g * GRU_H + 32 * grp + kk) * GRU_H + hid];

    float bsum[4] = {0.f, 0.f, 0.f, 0.f};          // column sums of dar, daz, dan, dgn over this block's rows and steps
    const int64_t ntiles = (a.rows + GRUB_BM - 1) / GRUB_BM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * GRUB_BM;
        const int last = (int)min((int64_t)GRUB_BM, a.rows - row0) - 1;
        float drec[GRUB_RT][4];
#pragma unroll
        for (int rt = 0; rt < GRUB_RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) drec[rt][i] = 0.f;

        for (int t = steps - 1; t >= 0; --t) {
            float(*gcur)[GRUB_PITCH] = gbuf[t & 1];
#pragma unroll
            for (int rt = 0; rt < GRUB_RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r_ = rt * 16 + grp * 4 + i;
                    const int64_t row = row0 + min(r_, last);
                    const float *gp = a.gates + (row * steps + t) * (4 * GRU_H) + hid;
                    const float r = gp[0], z = gp[GRU_H], n = gp[2 * GRU_H], q = gp[3 * GRU_H];
                    const float hprev = t > 0 ? a.hseq[(row * steps + t - 1) * GRU_H + hid] : 0.f;
                    float dh = drec[rt][i];
                    if (a.dh_seq) dh += a.dh_seq[(row * steps + t) * GRU_H + hid];
                    if (a.dh_sum) dh += a.dh_sum[row * GRU_H + hid];
                    const float dan = dh * (1.f - z) * (1.f - n * n);
                    const float daz = dh * (hprev - n) * z * (1.f - z);
                    const float dar = dan * q * r * (1.f - r);
                    const float dgn = dan * r;
                    drec[rt][i] = dh * z;                       // direct path; the W_hh path is added after the MFMAs
                    gcur[r_][hid] = dar;
                    gcur[r_][GRU_H + hid] = daz;
                    gcur[r_][2 * GRU_H + hid] = dgn;
                    if (r_ <= last) {
                        float *o = a.dgi + (row * steps + t) * (3 * GRU_H) + hid;
                        o[0] = dar; o[GRU_H] = daz; o[2 * GRU_H] = dan;
                        a.dghn[(row * steps + t) * GRU_H + hid] = dgn;
                        bsum[0] += dar; bsum[1] += daz; bsum[2] += dan; bsum[3] += dgn;
                    }
                }
            if (t == 0) break;                                   // h_{-1} is the constant 0: nothing to propagate
            __syncthreads();
#pragma unroll
            for (int rt = 0; rt < GRUB_RT; ++rt) {
                f4v acc0 = f4v{0.f, 0.f, 0.f, 0.f}, acc1 = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    float av[32];
                    const f4v *src = (const f4v *)(&gcur[rt * 16 + col][g * GRU_H + 32 * grp]);
#pragma unroll
                    for (int qd = 0; qd < 8; ++qd) {
                        const f4v v = src[qd];
                        av[4 * qd + 0] = v.x; av[4 * qd + 1] = v.y; av[4 * qd + 2] = v.z; av[4 * qd + 3] = v.w;
                    }
#pragma unroll
                    for (int kk = 0; kk < 32; kk += 2) {         // two accumulators: dependent MFMAs are 64 cycles apart
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], W[g][kk], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk + 1], W[g][kk + 1], acc1, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) drec[rt][i] += acc0[i] + acc1[i];
            }
        }
        __syncthreads();       // LDS is reused by the next tile
    }
    if (a.bias_partial) {      // the four lanes sharing a hidden unit (grp 0..3) hold disjoint rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bsum[i] += __shfl_xor(bsum[i], 16);
            bsum[i] += __shfl_xor(bsum[i], 32);
        }
        if (grp == 0) {
            float *o = a.bias_partial + (int64_t)blockIdx.x * (4 * GRU_H) + hid;
            o[0] = bsum[0]; o[GRU_H] = bsum[1]; o[2 * GRU_H] = bsum[2]; o[3 * GRU_H] = bsum[3];
        }
    }
}

// gru_seq_bwd_x3_kernel: the same backward recurrence with dGH·W_hh in split-bf16 arithmetic (see gru_proj_x3_kernel) and the
// transposed MFMA layout: a lane owns 4 consecutive hidden units of one row -> float4 loads of the saved gates / stores
// of dGI, bf16x4 stores of the split dGH rows.  32 rows per block iteration: the three dGH planes [32][384] are double
// buffered in 150 KB of LDS, one barrier per step.
constexpr int GBX_BM = 32;
constexpr int GBX_RT = GBX_BM / 16;
constexpr int GBX_PITCH = 3 * GRU_H + 8;      // bf16 elements

__global__ __launch_bounds__(512, 2) void gru_seq_bwd_x3_kernel(const GruBwdArgs a)
{
    __shared__ __bf16 Gs[2][3][GBX_BM][GBX_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int oc = wave * 16 + 4 * grp;       // the 4 hidden units this lane produces / consumes
    const int steps = a.steps;

    // MFMA A operand of the transposed product D[m = hidden j][n = row] = sum_k W_hh[k][j] dGH[row][k]:
    // A[m = 16w+col][k = c*32 + 8*grp + jj] = W_hh[k][16w + col]   (column gather, once per kernel)
    bf8v Wf[3][12];
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        float tmp[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tmp[jj] = a.whh[(int64_t)(c * 32 + 8 * grp + jj) * GRU_H + wave * 16 + col];
        bf16_split3_x8(tmp, Wf[0][c], Wf[1][c], Wf[2][c]);
    }
    const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};
    f4v bsum[4] = {zero4, zero4, zero4, zero4};    // column sums of dar, daz, dan, dgn over this block's rows and steps
    const int64_t ntiles = (a.rows + GBX_BM - 1) / GBX_BM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * GBX_BM;
        const int last = (int)min((int64_t)GBX_BM, a.rows - row0) - 1;
        f4v drec[GBX_RT];
#pragma unroll
        for (int rt = 0; rt < GBX_RT; ++rt) drec[rt] = zero4;

        for (int t = steps - 1; t >= 0; --t) {
            const int buf = t & 1;
#pragma unroll
            for (int rt = 0; rt < GBX_RT; ++rt) {
                const int r_ = rt * 16 + col;
                const int64_t e = (row0 + min(r_, last)) * steps + t;
                const float *gp = a.gates + e * (4 * GRU_H) + oc;
                const f4v r = *(const f4v *)gp, z = *(const f4v *)(gp + GRU_H), n = *(const f4v *)(gp + 2 * GRU_H), q = *(const f4v *)(gp + 3 * GRU_H);
                const f4v hprev = t > 0 ? *(const f4v *)(a.hseq + (e - 1) * GRU_H + oc) : zero4;
                f4v dh = drec[rt];
                if (a.dh_seq) dh += *(const f4v *)(a.dh_seq + e * GRU_H + oc);
                if (a.dh_sum) dh += *(const f4v *)(a.dh_sum + (row0 + min(r_, last)) * GRU_H + oc);
                const f4v dan = dh * (1.f - z) * (1.f - n * n);
                const f4v daz = dh * (hprev - n) * z * (1.f - z);
                const f4v dar = dan * q * r * (1.f - r);
                const f4v dgn = dan * r;
                drec[rt] = dh * z;                        // direct path; the W_hh path is added after the MFMAs
                auto publish = [&](int g, const f4v v) {
                    bf4v s0, s1, s2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        __bf16 x, y, w;
                        bf16_split3(v[j], x, y, w);
                        s0[j] = x; s1[j] = y; s2[j] = w;
                    }
                    *(bf4v *)(&Gs[buf][0][r_][g * GRU_H + oc]) = s0;
                    *(bf4v *)(&Gs[buf][1][r_][g * GRU_H + oc]) = s1;
                    *(bf4v *)(&Gs[buf][2][r_][g * GRU_H + oc]) = s2;
                };
                if (t > 0) { publish(0, dar); publish(1, daz); publish(2, dgn); }
                if (r_ <= last) {
                    float *o = a.dgi + e * (3 * GRU_H) + oc;
                    *(f4v *)o = dar; *(f4v *)(o + GRU_H) = daz; *(f4v *)(o + 2 * GRU_H) = dan;
                    *(f4v *)(a.dghn + e * GRU_H + oc) = dgn;
                    bsum[0] += dar; bsum[1] += daz; bsum[2] += dan; bsum[3] += dgn;
                }
            }
            if (t == 0) break;                            // h_{-1} is the constant 0: nothing to propagate
            __syncthreads();
#pragma unroll
            for (int rt = 0; rt < GBX_RT; ++rt) {
                f4v acc[3] = {zero4, zero4, zero4};
#pragma unroll
                for (int c = 0; c < 12; ++c) {
                    bf8v af[3];      // MFMA B operand: B[k][n = row] = dGH[row rt*16 + col][k = c*32 + 8*grp + j]
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) af[sp] = *(const bf8v *)(&Gs[buf][sp][rt * 16 + col][c * 32 + 8 * grp]);
                    // six partial products, smallest first, spread over three independent accumulators
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][c], af[2], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[2][c], af[0], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][c], af[1], acc[2], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][c], af[1], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][c], af[0], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][c], af[0], acc[2], 0, 0, 0);
                }
                drec[rt] += (acc[0] + acc[1]) + acc[2];
            }
        }
        __syncthreads();       // LDS is reused by the next tile
    }
    if (a.bias_partial) {      // the 16 lanes of a group (col 0..15) hold the 16 rows of a tile for the same 4 hidden units
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = bsum[g][j];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                bsum[g][j] = v;
            }
            if (col == 0) *(f4v *)(a.bias_partial + (int64_t)blockIdx.x * (4 * GRU_H) + g * GRU_H + oc) = bsum[g];
        }
    }
}

// gru_dx_x3_kernel: dX[rows, 128] = dGI[rows, 384] · W_ih   (gradient of the input projection w.r.t. its input), split-bf16
// arithmetic.  Same operand roles as gru_seq_bwd_x3_kernel's product (K = 384 gate columns, N = 128): the wave's column
// fragments of W_ih stay in 144 VGPRs, 32-row dGI tiles are split by the whole block into three bf16 LDS planes, double
// buffered with the next tile's global loads in flight during the MFMAs.  HBM-bound: 1536 B read + 512 B written per row.
struct DxArgs {
    int64_t rows;
    const float *g;       // [rows, 384]
    const float *w;       // [384, 128]
    float *out;           // [rows, ldo]
    int64_t ldo;
};

__global__ __launch_bounds__(512, 2) void gru_dx_x3_kernel(const DxArgs a)
{
    __shared__ __bf16 Gs[2][3][GBX_BM][GBX_PITCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int oc = wave * 16 + 4 * grp;

    bf8v Wf[3][12];       // A[m = 16w+col][k = c*32 + 8*grp + jj] = W[k][16w + col]
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        float tmp[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) tmp[jj] = a.w[(int64_t)(c * 32 + 8 * grp + jj) * GRU_H + wave * 16 + col];
        bf16_split3_x8(tmp, Wf[0][c], Wf[1][c], Wf[2][c]);
    }
    const int64_t ntiles = (a.rows + GBX_BM - 1) / GBX_BM;
    // staging role: 32 rows x 96 float4 = 3072 float4, six per thread; idx -> (row, c4)
    auto load_tile = [&](int64_t tile, f4v (&v)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = tid + 512 * i;
            int64_t r = tile * GBX_BM + idx / 96;
            r = r < a.rows ? r : a.rows - 1;
            v[i] = __builtin_nontemporal_load((const f4v *)(a.g + r * (3 * GRU_H) + (idx % 96) * 4));
        }
    };
    auto stage_tile = [&](int buf, const f4v (&v)[6]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int idx = tid + 512 * i;
            bf4v s0, s1, s2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                __bf16 p, q, r;
                bf16_split3(v[i][j], p, q, r);
                s0[j] = p; s1[j] = q; s2[j] = r;
            }
            __bf16 *dst = &Gs[buf][0][idx / 96][(idx % 96) * 4];
            *(bf4v *)dst = s0;
            *(bf4v *)(dst + GBX_BM * GBX_PITCH) = s1;
            *(bf4v *)(dst + 2 * GBX_BM * GBX_PITCH) = s2;
        }
    };
    f4v stage[6];
    int buf = 0;
    if ((int64_t)blockIdx.x < ntiles) {
        load_tile(blockIdx.x, stage);
        stage_tile(0, stage);
    }
    __syncthreads();
    const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int64_t next = tile + gridDim.x;
        if (next < ntiles) load_tile(next, stage);
        const int64_t row0 = tile * GBX_BM;
#pragma unroll
        for (int rt = 0; rt < GBX_RT; ++rt) {
            f4v acc[3] = {zero4, zero4, zero4};
#pragma unroll
            for (int c = 0; c < 12; ++c) {
                bf8v af[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) af[sp] = *(const bf8v *)(&Gs[buf][sp][rt * 16 + col][c * 32 + 8 * grp]);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][c], af[2], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[2][c], af[0], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][c], af[1], acc[2], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][c], af[1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[1][c], af[0], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[0][c], af[0], acc[2], 0, 0, 0);
            }
            const int64_t row = row0 + rt * 16 + col;
            if (row < a.rows) *(f4v *)(a.out + row * a.ldo + oc) = (acc[0] + acc[1]) + acc[2];
        }
        if (next < ntiles) stage_tile(buf ^ 1, stage);
        __syncthreads();
    }
}

// gru_dw_x3_kernel: weight gradients  dW[384, 128] = Σ_r G[r, :]ᵀ · X[r, :]  over R = rows*steps row-steps (dW_ih: G = dGI,
// X = the layer input; dW_hh: G = [dGI_r, dGI_z, dGHn], X = h_{t-1}), split-bf16 arithmetic.  The reduction index is the
// ROW, so both operands are staged TRANSPOSED: a lane loads 4 consecutive rows of one column (coalesced across the
// wave), splits them and stores bf16x4 into planes [column][row]; MFMA fragments are then single ds_read_b128.
// A block owns 192 of the 384 gate columns (LDS: 2 buffers x 3 planes x (192 + 128) x 40 bf16 = 150 KB); blocks b and
// b + 8 (same XCD, shared L2) walk the same rows with the two halves, so X is fetched from HBM once.  Every block pair
// adds its partial sum to `partial[pair]` (its own slice, no atomics); the host adds the pairs (deterministic).
// shift: X row r is taken as (r % steps ? X[r - 1] : 0) — h_{t-1} straight from the saved h sequence, no shifted copy.
// Measured 0.40 ms per 0.5 M row-steps (hipBLASLt fp32 split-K TN GEMM: 0.50 ms).  Ablation (remove one component, keep the
// rest): no MFMA 0.28, no global loads 0.28, no staging 0.28, no barrier 0.37 — the three costs (~0.12 ms each) ADD even
// with the loads three chunks ahead and the staging VALU interleaved between the MFMAs: on a SIMD the matrix pipe and
// the VALU do not run concurrently for this instruction mix, and the 160 dword-per-lane buffer loads of a chunk keep the
// CU's address unit busy for about as long as the MFMAs take.  Next: 16-byte loads + an in-register 4x4 transpose
// (4x fewer address-unit cycles), fewer VALU ops per value in the split.
constexpr int DW_KB = 32;          // rows per chunk = MFMA K
constexpr int DW_MH = 192;         // gate columns per block
constexpr int DW_PITCH = 40;       // bf16 per transposed plane row: 32 + 8 pad, 80-byte rows stay 16-byte aligned

struct DwArgs {
    int64_t rows;          // R
    int32_t steps;
    int32_t shift;
    const float *g01;      // gate columns [0, 256)
    int64_t ldg01;
    const float *g2;       // gate columns [256, 384)
    int64_t ldg2;
    const float *x;        // [R, 128]
    int64_t ldx;
    float *partial;        // [pairs, 384, 128]
    int32_t pairs;
    int32_t accumulate;    // partial += instead of partial =
};

__global__ __launch_bounds__(512, 2) void gru_dw_x3_kernel(const DwArgs a)
{
    __shared__ __bf16 Gt[2][3][DW_MH][DW_PITCH];
    __shared__ __bf16 Xt[2][3][GRU_H][DW_PITCH];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int col = lane & 15, grp = lane >> 4;
    const int pair = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7), half = (blockIdx.x >> 3) & 1;
    const int mg = wv >> 1, ng = wv & 1;          // wave tile: m-tiles 3mg..3mg+2 (of 12), n-tiles 4ng..4ng+3 (of 8)

    const int64_t nchunks = (a.rows + DW_KB - 1) / DW_KB;
    const int64_t per = (nchunks + a.pairs - 1) / a.pairs;
    const int64_t c_lo = min((int64_t)pair * per, nchunks), c_hi = min(c_lo + per, nchunks);
    const int sh = a.shift ? 1 : 0;

    // Staging: a wave-item is 32 columns x 2 strips (of 4 rows): lane -> column lc = (lane & 7) + 8*(lane >> 4), strip parity
    // par = bit 3 of the lane.  The 16 lanes that store in one LDS clock are then 8 columns x both strips of a 16-byte
    // slot: with 80-byte plane rows (20 dwords: 8 consecutive columns start on the 8 different bank quads) the bf16x4
    // stores are conflict-free.  [A lane -> 64 consecutive columns mapping puts lanes l and l+8 on the same bank pair:
    // 8-way conflicts that made staging the most expensive phase of the kernel.]
    // G half = 4 strip pairs x 6 column groups = 24 wave-items, X = 4 x 4 = 16: five per wave (3 of G, 2 of X).  Item
    // geometry, source matrix, rows and the step shift are wave-uniform (scalar unit); the per-lane part of an address is a
    // constant offset.  Loads are unconditional and nothing touches the loaded registers until stage_chunk (a guarded
    // load becomes its own basic block behind a s_waitcnt; a select right after the load would wait for it).
    // A wave issues at most one instruction every ~4 cycles, so the 72 MFMAs of a chunk (1152 cycles) leave room for
    // only ~300 other instructions per wave and chunk.  The first versions spent ~2000 (64-bit scalar address products,
    // a 64-bit modulo, clamps for every load) and ran at hipBLASLt's fp32 speed no matter what else was tuned.  Hence
    // buffer loads: one descriptor per item (base = the item's first column at the block's first row, range = up to the
    // item's columns in the last valid row), a per-lane byte offset that advances by one v_add per load and chunk, and
    // the hardware range check returns 0 for rows past the end (and for row -1 of the shifted operand, whose offset
    // wraps; that row is a t = 0 row) — no clamps, no tail masks.  Descriptor inputs go through readfirstlane so that the compiler can prove
    // them wave-uniform (otherwise every load is wrapped in a waterfall loop).
    const int lc = (lane & 7) + 8 * (lane >> 4), par = (lane >> 3) & 1;
    const int64_t row_lo = c_lo * DW_KB;
    __amdgpu_buffer_rsrc_t rs_[5];
    int sp_[5], cb_[5];                 // item geometry (scalar): strip pair, first column
    uint32_t voff_[5][4];               // per lane byte offset of row (8*sp + 4*par + j [- shift]) of the current chunk, column lc
    uint32_t adv_[5];                   // bytes per chunk
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = 8 * (i < 3 ? i : i - 3) + wv;
        int64_t ld, last_row, first_row; // rows of the matrix this item's descriptor spans
        int back = 0;                    // rows between the descriptor base and the block's first row
        const float *src;
        if (i < 3) {
            sp_[i] = q / 6; cb_[i] = 32 * (q % 6);
            const int m0 = half * DW_MH + cb_[i];                        // 32 columns m0 .. m0+31 come from one matrix
            ld = m0 < 2 * GRU_H ? a.ldg01 : a.ldg2;
            src = m0 < 2 * GRU_H ? a.g01 + m0 : a.g2 + (m0 - 2 * GRU_H);
            last_row = a.rows - 1;
        } else {
            sp_[i] = q >> 2; cb_[i] = 32 * (q & 3);
            ld = a.ldx;
            src = a.x + cb_[i];
            last_row = a.rows - 1 - sh;                                  // row r reads x[r - 1] ...
            back = row_lo >= sh ? sh : 0;                                // ... so the span starts one row before the block's
        }                                                                // (row -1 of the whole matrix: offset wraps -> 0)
        first_row = row_lo - back;
        const uint64_t base = (uint64_t)(src + first_row * ld);
        const int64_t bytes = (last_row - first_row) * ld * 4 + 32 * 4;  // may be <= 0: nothing readable
        const uint32_t b_lo = __builtin_amdgcn_readfirstlane((uint32_t)base), b_hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)max(min(bytes, (int64_t)0x7fffffff), (int64_t)0));
        rs_[i] = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)b_hi << 32) | b_lo), 0, nrec, 0x00020000);
        adv_[i] = (uint32_t)(DW_KB * ld * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            voff_[i][j] = (uint32_t)(((8 * sp_[i] + 4 * par + j - (i < 3 ? 0 : sh) + back) * ld + lc) * 4);
    }
    // Loads must be issued for consecutive chunks (the offsets advance by one chunk per call); nothing touches the loaded
    // registers until stage_chunk (a select right after a load would wait for it before the MFMAs instead of after).
    auto load_chunk = [&](f4v (&v)[5]) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_[i], voff_[i][j], 0, 0));
                voff_[i][j] += adv_[i];
            }
    };
    auto split4 = [&](const f4v v, __bf16 *dst, int plane_elems) {
        bf4v s0, s1, s2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __bf16 p, q, r;
            bf16_split3(v[j], p, q, r);
            s0[j] = p; s1[j] = q; s2[j] = r;
        }
        *(bf4v *)dst = s0;
        *(bf4v *)(dst + plane_elems) = s1;
        *(bf4v *)(dst + 2 * plane_elems) = s2;
    };
    // Step bookkeeping for the shifted operand, branch-free: t = step index (mod steps) of a row; rows with t == 0 are h_{-1} = 0.
    // Without shift the same code runs with steps = INT_MAX and t starting at 1 (never 0), so that a whole loop iteration
    // is ONE basic block and the scheduler can slot the split/store VALU work between the MFMAs (3 issue slots per MFMA).
    const int steps_e = a.shift ? a.steps : 0x7fffffff;
    int toff_[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        toff_[i][0] = a.shift ? (8 * sp_[3 + i]) % a.steps : 0;
        toff_[i][1] = a.shift ? (8 * sp_[3 + i] + 4) % a.steps : 0;
    }
    const int tadv = a.shift ? DW_KB % a.steps : 0;
    auto wrap = [&](int t) { return t >= steps_e ? t - steps_e : t; };
    // stage one of the five items of a chunk: split + three bf16x4 stores; the t = 0 rows of the shifted operand are zeroed
    // here, rows past the end already arrived as 0
    auto stage_item = [&](int i, int buf, int tbase, const f4v (&vv)[5]) {
        f4v v = vv[i];
        if (i >= 3) {
            int t0 = wrap(tbase + toff_[i - 3][0]), t1 = wrap(tbase + toff_[i - 3][1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool keep = par ? t1 != 0 : t0 != 0;
                v[j] = keep ? v[j] : 0.f;
                t0 = (t0 + 1 == steps_e) ? 0 : t0 + 1;
                t1 = (t1 + 1 == steps_e) ? 0 : t1 + 1;
            }
        }
        if (i < 3) split4(v, &Gt[buf][0][cb_[i] + lc][8 * sp_[i] + 4 * par], DW_MH * DW_PITCH);
        else split4(v, &Xt[buf][0][cb_[i] + lc][8 * sp_[i] + 4 * par], GRU_H * DW_PITCH);
    };

    const f4v zero4 = f4v{0.f, 0.f, 0.f, 0.f};
    f4v acc[3][4];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = zero4;

    // One iteration: LDS buffer `buf` holds chunk `chunk`; `nxt` registers hold chunk+1; chunk+3 is loaded into `far` (the
    // set that held `chunk`, already staged).  No guards: loads past the block's range are out of range for the
    // descriptors (no memory access, zeros), and staging a chunk nobody multiplies is harmless.
    int tb = a.shift ? (int)((c_lo * DW_KB) % a.steps) : 1;              // step index of the first row of the chunk being staged
    auto iteration = [&](int buf, f4v (&far)[5], const f4v (&nxt)[5]) {
        load_chunk(far);
        tb = wrap(tb + tadv);
        bf8v af[3][3];     // [m tile][split]: A[m = gate column][k = row 8*grp + j]
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) af[mt][sp] = *(const bf8v *)(&Gt[buf][sp][(3 * mg + mt) * 16 + col][8 * grp]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bf8v bfr[3];   // B[k = row][n = input column]
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) bfr[sp] = *(const bf8v *)(&Xt[buf][sp][(4 * ng + nt) * 16 + col][8 * grp]);
#define CTGCN_X3_MFMA(I, J)                                                                                              \
            _Pragma("unroll") for (int mt = 0; mt < 3; ++mt)                                                             \
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][I], bfr[J], acc[mt][nt], 0, 0, 0);
            CTGCN_X3_PAIRS(CTGCN_X3_MFMA)
#undef CTGCN_X3_MFMA
            stage_item(nt, buf ^ 1, tb, nxt);                          // the next chunk's staging rides under these MFMAs
        }
        stage_item(4, buf ^ 1, tb, nxt);
        __syncthreads();
    };

    f4v va[5], vb[5], vc[5];           // three register sets: loads run three chunks ahead of their use
    load_chunk(va);
    load_chunk(vb);
    load_chunk(vc);
    if (c_lo < c_hi) {
#pragma unroll
        for (int i = 0; i < 5; ++i) stage_item(i, 0, tb, va);
    }
    __syncthreads();
    for (int64_t chunk = c_lo; chunk < c_hi; chunk += 6) {              // buffer parity and register set both repeat after 6
        iteration(0, va, vb);                                          // chunk+1 sits in b, chunk+2 in c, chunk+3 goes to a
        if (chunk + 1 < c_hi) iteration(1, vb, vc);
        if (chunk + 2 < c_hi) iteration(0, vc, va);
        if (chunk + 3 < c_hi) iteration(1, va, vb);
        if (chunk + 4 < c_hi) iteration(0, vb, vc);
        if (chunk + 5 < c_hi) iteration(1, vc, va);
    }
    // D layout: lane (col, grp) holds D[m = 4*grp + i][n = col].  All 48 reads of the running partial sums are issued
    // before the first add (one guarded read-modify-write per element serialises 48 memory latencies per call).
    float *out = a.partial + (int64_t)pair * (3 * GRU_H) * GRU_H;
    auto oidx = [&](int mt, int nt, int i) { return (half * DW_MH + (3 * mg + mt) * 16 + 4 * grp + i) * GRU_H + (4 * ng + nt) * 16 + col; };
    if (a.accumulate) {
        f4v prev[3][4];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) prev[mt][nt][i] = out[oidx(mt, nt, i)];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] += prev[mt][nt];
    }
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[oidx(mt, nt, i)] = acc[mt][nt][i];
}

// transpose_bias_kernel: out[i][j] = w[j][i] + bias[j] — Linear applied to one-hot (identity) node features is just Wᵀ + b
// (reference helper.py:161-172 builds the identity, layers.py:95-106 multiplies by it).  One 64 x 64 tile per block through
// LDS: 256-byte reads along i, 256-byte writes along j, 16 bytes per lane both ways (VEC; otherwise element by element at
// the edges and for unaligned rows).  HBM-bound: 4 B read + 4 B written per element.
struct TransposeGroup { const float *w, *bias; float *out; };
template <bool VEC>
__device__ __forceinline__ void transpose_bias_body(int64_t n, int d, const float *__restrict__ w, int64_t ldw,
                                                    const float *__restrict__ bias, float *__restrict__ out, int64_t ldo)
{
    __shared__ float tile[64][65];                        // [j][i]
    const int tid = threadIdx.x, r = tid >> 4, c4 = (tid & 15) * 4;
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int j0 = blockIdx.y * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = r + 16 * k;
        f4v v = f4v{0.f, 0.f, 0.f, 0.f};
        if (j0 + j < d) {
            const float *src = w + (int64_t)(j0 + j) * ldw + i0 + c4;
            if (VEC && i0 + c4 + 3 < n) v = *(const f4v *)src;
            else
                for (int q = 0; q < 4; ++q) if (i0 + c4 + q < n) v[q] = src[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[j][c4 + q] = v[q];
    }
    __syncthreads();
    f4v b = f4v{0.f, 0.f, 0.f, 0.f};
    if (bias)
        for (int q = 0; q < 4; ++q) if (j0 + c4 + q < d) b[q] = bias[j0 + c4 + q];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = r + 16 * k;
        if (i0 + i >= n) continue;
        f4v v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = tile[c4 + q][i] + b[q];
        float *dst = out + (i0 + i) * ldo + j0 + c4;
        if (VEC && j0 + c4 + 3 < d) __builtin_nontemporal_store(v, (f4v *)dst);
        else
            for (int q = 0; q < 4; ++q) if (j0 + c4 + q < d) dst[q] = v[q];
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_bias_kernel(int64_t n, int d, const float *__restrict__ w, int64_t ldw,
                                                             const float *__restrict__ bias, float *__restrict__ out, int64_t ldo)
{
    transpose_bias_body<VEC>(n, d, w, ldw, bias, out, ldo);
}
// Linear(I) of every snapshot of a window in one launch: blockIdx.z = snapshot (its own weight, models.py:225-227)
template <bool VEC>
__global__ __launch_bounds__(256) void transpose_bias_group_kernel(int64_t n, int d, const TransposeGroup *__restrict__ table, int64_t ldw, int64_t ldo)
{
    const TransposeGroup G = table[blockIdx.z];
    transpose_bias_body<VEC>(n, d, G.w, ldw, G.bias, G.out, ldo);
}

// final core numbers; with a level cap the unpeeled vertices (current degree >= cap) are reported as `cap`
__global__ void kcore_copy_kernel(int n, int cap, const int32_t *__restrict__ deg, int32_t *__restrict__ core)
{
    const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (v < n) core[v] = min(deg[v], cap);
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

// =================================================================================== C ABI
extern "C" size_t ctgcn_ingest_workspace_bytes_(int64_t n, int64_t m);   // ctgcn_ingest.hip

// shared with the other translation units of the library (not part of the public header)
extern "C" int ctgcn_set_error_(int code, const char *msg) { return fail(code, "%s", msg); }
extern "C" void ctgcn_set_persistent_cus(int cus) { g_persistent_cus = cus; }
extern "C" int ctgcn_persistent_cus_(int device_cus) { return persistent_cus(device_cus); }      // ctgcn_gru_bwd.hip sizes its grids with it
extern "C" int ctgcn_split_rows_mapped_(int64_t rows, int32_t k, int32_t kp, const float *x, int64_t ldx, void *p1, void *p2, float *scale,
                                        const int32_t *group_map, int32_t group, float residual_scale, void *stream);   // ctgcn_gemm.hip

extern "C" {

int ctgcn_abi_version(void) { return CTGCN_ABI_VERSION; }

// diagnostic counters of the grouped launches' descriptor tables (ctgcn_table.h): tables written / found current through their shadow
static std::atomic<uint64_t> g_table_written{0}, g_table_current{0};
void ctgcn_table_count_(int written) { (written ? g_table_written : g_table_current).fetch_add(1, std::memory_order_relaxed); }
uint64_t ctgcn_table_uploads(int current) { return (current ? g_table_current : g_table_written).load(std::memory_order_relaxed); }

const char *ctgcn_last_error(void) { return g_err; }

int ctgcn_device_info(char *name_host, size_t name_len, int *cu_count_host)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (name_host && name_len) snprintf(name_host, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
    return CTGCN_OK;
}

int ctgcn_transpose_bias_f32(int64_t n, int32_t d, const float *w, int64_t ldw, const float *bias, float *out, int64_t ldo,
                             void *stream)
{
    if (n < 0 || d < 1 || ldw < n || ldo < d) return fail(CTGCN_E_INVALID, "transpose_bias: bad sizes n=%lld d=%d", (long long)n, d);
    if (n == 0) return CTGCN_OK;
    if (!w || !out) return fail(CTGCN_E_INVALID, "transpose_bias: null pointer");
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((d + 63) / 64));
    if (!(ldw & 3) && !(ldo & 3) && aligned16(w) && aligned16(out))
        hipLaunchKernelGGL(transpose_bias_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, n, d, w, ldw, bias, out, ldo);
    else
        hipLaunchKernelGGL(transpose_bias_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, n, d, w, ldw, bias, out, ldo);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_spmm_csr_f32(int64_t n_rows, int32_t d, const int32_t *row_ptr, const int32_t *col_idx,
                       const float *val, const float *X, int64_t ldx, float *Y, int64_t ldy,
                       int accumulate, void *stream)
{
    if (n_rows < 0 || d <= 0 || ldx < d || ldy < d) return fail(CTGCN_E_INVALID, "spmm_csr: bad sizes n=%lld d=%d ldx=%lld ldy=%lld", (long long)n_rows, d, (long long)ldx, (long long)ldy);
    if (n_rows == 0) return CTGCN_OK;
    if (!row_ptr || !X || !Y) return fail(CTGCN_E_INVALID, "spmm_csr: null pointer");
    AggArgs a{};
    a.n = n_rows; a.d = d; a.K = 1;
    a.row_ptr = row_ptr; a.col = col_idx; a.val = val; a.slot = nullptr;
    a.src = X; a.ldsrc = ldx; a.self = nullptr; a.out = Y; a.out_ld = ldy;
    a.flags = CTGCN_F_NESTED; a.accumulate = accumulate ? 1 : 0;
    const bool v4 = (d % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && aligned16(X) && aligned16(Y);
    return launch_agg<true>(a, v4, (hipStream_t)stream);
}

int ctgcn_core_aggregate_f32(int64_t n_rows, int32_t d, int32_t K, const int32_t *row_ptr,
                             const int32_t *col_idx, const float *val, const uint8_t *slot,
                             const float *X, int64_t ldx, float *H, uint32_t flags,
                             const int32_t *long_rows, int32_t n_long, int32_t long_threshold,
                             int32_t hub_split, void *hub_workspace, size_t hub_workspace_bytes, void *stream)
{
    if (n_rows < 0 || d <= 0 || ldx < d) return fail(CTGCN_E_INVALID, "core_aggregate: bad sizes n=%lld d=%d ldx=%lld", (long long)n_rows, d, (long long)ldx);
    if (K < 1 || K > CTGCN_MAX_SLOTS) return fail(CTGCN_E_INVALID, "core_aggregate: K=%d outside [1,%d]", K, CTGCN_MAX_SLOTS);
    if (n_rows == 0) return CTGCN_OK;
    if (!row_ptr || !X || !H) return fail(CTGCN_E_INVALID, "core_aggregate: null pointer");
    if (!slot && K != 1) return fail(CTGCN_E_INVALID, "core_aggregate: slot tags required when K > 1");
    AggArgs a{};
    a.n = n_rows; a.d = d; a.K = K;
    a.row_ptr = row_ptr; a.col = col_idx; a.val = val; a.slot = slot;
    a.src = X; a.ldsrc = ldx; a.self = nullptr; a.out = H; a.out_ld = (int64_t)K * d;
    a.flags = flags; a.accumulate = 0;
    a.long_rows = long_rows; a.n_long = n_long; a.long_thresh = long_threshold;
    if (n_long > 0 && long_threshold < 1) return fail(CTGCN_E_INVALID, "core_aggregate: long_threshold must be >= 1");
    if (int rc = set_hub_pieces(a, K, hub_split, hub_workspace, hub_workspace_bytes, "core_aggregate")) return rc;
    const bool v4 = (d % 4 == 0) && (ldx % 4 == 0) && aligned16(X) && aligned16(H);
    return launch_agg<true>(a, v4, (hipStream_t)stream);
}

size_t ctgcn_core_aggregate_split_workspace_bytes(int64_t n_rows, int32_t d, int32_t K, int32_t n_out, int32_t n_long)
{
    if (n_rows < 0 || d < 1 || K < 1 || n_out < 1 || n_long < 0) return 0;
    // (n_rows rounded up to whole tiles of 64: the compact operand rows of a row plan include the padding of the last tile)
    return ctgcn_linear_workspace_bytes((n_rows + 63) / 64 * 64 * K, n_out, d) + ((size_t)n_long * K * d * 4 + 255) / 256 * 256;
}

int ctgcn_core_aggregate_split_f32(int64_t n_rows, int32_t d, int32_t K, const int32_t *row_ptr, const int32_t *col_idx,
                                   const float *val, const uint8_t *slot, const float *X, int64_t ldx, uint32_t flags,
                                   const int32_t *long_rows, int32_t n_long, int32_t long_threshold, int32_t n_out,
                                   const int32_t *row_order, const uint32_t *tile_mask, const int32_t *tile_base, int64_t operand_rows,
                                   const int32_t *hub_row_dest,
                                   int32_t hub_split, void *hub_workspace, size_t hub_workspace_bytes,
                                   void *workspace, size_t workspace_bytes, void *stream)
{
    if (n_rows < 0 || d <= 0 || ldx < d || n_out < 1) return fail(CTGCN_E_INVALID, "core_aggregate_split: bad sizes n=%lld d=%d ldx=%lld", (long long)n_rows, d, (long long)ldx);
    if (K < 1 || K > CTGCN_MAX_SLOTS) return fail(CTGCN_E_INVALID, "core_aggregate_split: K=%d outside [1,%d]", K, CTGCN_MAX_SLOTS);
    if (n_rows == 0) return CTGCN_OK;
    if (!row_ptr || !X || !workspace) return fail(CTGCN_E_INVALID, "core_aggregate_split: null pointer");
    if (!slot && K != 1) return fail(CTGCN_E_INVALID, "core_aggregate_split: slot tags required when K > 1");
    if ((d & 3) || d > 512 || (ldx & 3) || !aligned16(X) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail(CTGCN_E_UNSUPPORTED, "core_aggregate_split: needs d %% 4 == 0, d <= 512, 16-byte aligned rows, 256-byte aligned workspace");
    if (n_long < 0 || (n_long > 0 && (!long_rows || long_threshold < 1))) return fail(CTGCN_E_INVALID, "core_aggregate_split: bad hub row list");
    if ((row_order == nullptr) != (tile_mask == nullptr)) return fail(CTGCN_E_INVALID, "core_aggregate_split: row_order and tile_mask come together");
    if (row_order && K > 64) return fail(CTGCN_E_UNSUPPORTED, "core_aggregate_split: a row plan needs K <= 64");
    if (row_order && ((d == 128 && n_out == 1) != (tile_base == nullptr)))
        return fail(CTGCN_E_INVALID, "core_aggregate_split: tile_base (compact operand rows, tiles of 64) goes with the GEMM consumer, tiles of 16 without it with the GRU layer kernel");
    if (!row_order && tile_base) return fail(CTGCN_E_INVALID, "core_aggregate_split: tile_base without a row plan");
    if (tile_base && (operand_rows < 1 || operand_rows > (n_rows + 63) / 64 * 64 * K))
        return fail(CTGCN_E_INVALID, "core_aggregate_split: operand_rows=%lld outside [1, ceil64(n_rows) K]", (long long)operand_rows);
    if (row_order && n_long > 0 && !hub_row_dest) return fail(CTGCN_E_INVALID, "core_aggregate_split: hub rows under a row plan need their destination rows");
    if (workspace_bytes < ctgcn_core_aggregate_split_workspace_bytes(n_rows, d, K, n_out, n_long))
        return fail(CTGCN_E_WORKSPACE, "core_aggregate_split: workspace too small (ctgcn_core_aggregate_split_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = tile_base ? operand_rows : n_rows * K;     // operand rows of the consumer: the planes' layout follows ITS row count
    const int32_t kp = (d + 63) / 64 * 64;                // the k padding of ctgcn_linear_f32
    _Float16 *p1 = (_Float16 *)workspace, *p2 = p1 + (size_t)rows * kp;
    float *scale = (float *)(p2 + (size_t)rows * kp);
    float *hub = (float *)((char *)workspace + ctgcn_linear_workspace_bytes((n_rows + 63) / 64 * 64 * K, n_out, d));
    AggArgs a{};
    a.n = n_rows; a.d = d; a.K = K;
    a.row_ptr = row_ptr; a.col = col_idx; a.val = val; a.slot = slot;
    a.src = X; a.ldsrc = ldx; a.self = nullptr; a.out = hub; a.out_ld = (int64_t)K * d;
    a.flags = flags; a.accumulate = 0;
    a.long_rows = long_rows; a.n_long = n_long; a.long_thresh = long_threshold;
    a.order = row_order; a.tmask = tile_mask; a.tbase = tile_base; a.tile_shift = tile_base ? 6 : 4;
    const AggPlan p = plan_for(d, true);
    a.chunks = p.chunks;
    a.passes = p.passes;
    if (a.n_long > 0 && (size_t)K * p.lpr * 16 > HUB_LDS_BUDGET) a.n_long = 0;      // K*d too large for the hub kernel's LDS partials
    if (a.n_long <= 0) { a.n_long = 0; a.long_thresh = 0x7fffffff; }
    if (int rc = set_hub_pieces(a, K, hub_split, hub_workspace, hub_workspace_bytes, "core_aggregate_split")) return rc;
    const int rows_per_block = p.chunks <= 32 ? 8 : 4;
    const int64_t blocks = (n_rows + rows_per_block - 1) / rows_per_block;
    if (blocks > 0x7fffffffLL) return fail(CTGCN_E_UNSUPPORTED, "core_aggregate_split: grid too large");
    // the residual plane is unscaled for every consumer (GRU layer kernel and GEMM both add the three products in one accumulator)
    const float rsc = 1.f;
    static const int force_u8 = [] { const char *e = getenv("CTGCN_AGG_U8"); return e ? atoi(e) : -1; }();      // A/B: 1 = always eight gathers in flight, 0 = never
    const bool u8 = force_u8 >= 0 ? force_u8 != 0 : n_rows <= 200000;
    if (row_order && K > 32) {        // a row plan over 33-64 slots (two mask words per tile): lists this deep belong to graphs of ~1 000 nodes
        if (p.chunks <= 32) hipLaunchKernelGGL((agg_fwd_split_kernel<32, 1, 8, true>), dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
        else if (p.chunks <= 64) hipLaunchKernelGGL((agg_fwd_split_kernel<64, 1, 8, true>), dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
        else hipLaunchKernelGGL((agg_fwd_split_kernel<64, 2, 8, true>), dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
    } else
    if (p.chunks <= 32 && u8) hipLaunchKernelGGL(agg_fwd_split32_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
    else if (p.chunks <= 32) hipLaunchKernelGGL(agg_fwd_split32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
    else if (p.chunks <= 64) hipLaunchKernelGGL((agg_fwd_split_kernel<64, 1, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
    else if (n_rows <= 200000) hipLaunchKernelGGL((agg_fwd_split_kernel<64, 2, 8>), dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
    else hipLaunchKernelGGL((agg_fwd_split_kernel<64, 2, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a, p1, p2, scale, kp, rsc);
    HIP_TRY(hipGetLastError());
    if (a.n_long > 0) {
        // hub rows: the block-per-row kernel writes fp32 rows into the compact scratch, then they are split like any other rows
        a.hub_compact = 1;
        const size_t per_group = (size_t)K * p.lpr * 16;
        int G = HUB_THREADS / p.lpr;
        while (G > 1 && per_group * G > HUB_LDS_BUDGET) --G;
        a.hub_groups = G;
        const size_t lds = per_group * G;
#define HUBCASE(L)                                                                                                                        \
        if (p.lpr == L) {                                                                                                                 \
            auto k = agg_fwd_hub_kernel<4, L, 4>;                                                                                        \
            if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(k, dim3((unsigned)a.n_long * (unsigned)(a.hub_part ? a.hub_split : 1)), dim3(HUB_THREADS), lds, st, a);    \
        }
        HUBCASE(8) else HUBCASE(16) else HUBCASE(32) else HUBCASE(64)
#undef HUBCASE
        if (a.hub_part) hipLaunchKernelGGL((agg_hub_final_kernel<4, true>), dim3((unsigned)a.n_long), dim3(256), 0, st, a);
        HIP_TRY(hipGetLastError());
        const int rc = row_order ? ctgcn_split_rows_mapped_((int64_t)a.n_long * K, d, kp, hub, d, p1, p2, scale, hub_row_dest, -1, rsc, stream)
                                 : ctgcn_split_rows_mapped_((int64_t)a.n_long * K, d, kp, hub, d, p1, p2, scale, long_rows, K, rsc, stream);
        if (rc != CTGCN_OK) return rc;
    }
    return CTGCN_OK;
}

int ctgcn_core_aggregate_bwd_prep_f32(int64_t n_rows, int32_t d, int32_t K, const float *dH,
                                      const float *H, float *Z, float *S0, uint32_t flags, void *stream)
{
    if (n_rows < 0 || d <= 0 || K < 1 || K > CTGCN_MAX_SLOTS) return fail(CTGCN_E_INVALID, "bwd_prep: bad sizes");
    if (n_rows == 0) return CTGCN_OK;
    if (!dH || !Z || ((flags & CTGCN_F_RELU) && !H)) return fail(CTGCN_E_INVALID, "bwd_prep: null pointer");
    const bool v4 = (d % 4 == 0) && aligned16(dH) && aligned16(Z) && (!H || aligned16(H)) && (!S0 || aligned16(S0));
    const int vec = v4 ? 4 : 1;
    const int chunks = (d + vec - 1) / vec;
    const int64_t total = n_rows * chunks;
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(CTGCN_E_UNSUPPORTED, "bwd_prep: grid too large");
    if (v4)
        hipLaunchKernelGGL(agg_bwd_prep_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_rows, d, K, chunks, dH, H, Z, S0, flags);
    else
        hipLaunchKernelGGL(agg_bwd_prep_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_rows, d, K, chunks, dH, H, Z, S0, flags);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_core_aggregate_bwd_f32(int64_t n_rows, int32_t d, int32_t K, const int32_t *row_ptr,
                                 const int32_t *col_idx, const float *val, const uint8_t *slot,
                                 const float *Z, const float *S0, float *dX, int64_t lddx,
                                 uint32_t flags, const int32_t *long_rows, int32_t n_long,
                                 int32_t long_threshold, int32_t hub_split, void *hub_workspace, size_t hub_workspace_bytes, void *stream)
{
    if (n_rows < 0 || d <= 0 || lddx < d || K < 1 || K > CTGCN_MAX_SLOTS) return fail(CTGCN_E_INVALID, "core_aggregate_bwd: bad sizes");
    if (n_rows == 0) return CTGCN_OK;
    if (!row_ptr || !Z || !dX) return fail(CTGCN_E_INVALID, "core_aggregate_bwd: null pointer");
    if (!slot && K != 1) return fail(CTGCN_E_INVALID, "core_aggregate_bwd: slot tags required when K > 1");
    if ((flags & CTGCN_F_SELF_LOOP) && !S0) return fail(CTGCN_E_INVALID, "core_aggregate_bwd: S0 required with SELF_LOOP");
    AggArgs a{};
    a.n = n_rows; a.d = d; a.K = K;
    a.row_ptr = row_ptr; a.col = col_idx; a.val = val; a.slot = slot;
    a.src = Z; a.ldsrc = 0; a.self = (flags & CTGCN_F_SELF_LOOP) ? S0 : nullptr;
    a.out = dX; a.out_ld = lddx; a.flags = flags; a.accumulate = 0;
    a.long_rows = long_rows; a.n_long = n_long; a.long_thresh = long_threshold;
    if (n_long > 0 && long_threshold < 1) return fail(CTGCN_E_INVALID, "core_aggregate_bwd: long_threshold must be >= 1");
    if (int rc = set_hub_pieces(a, 1, hub_split, hub_workspace, hub_workspace_bytes, "core_aggregate_bwd")) return rc;
    const bool v4 = (d % 4 == 0) && (lddx % 4 == 0) && aligned16(Z) && aligned16(dX) && (!a.self || aligned16(a.self));
    return launch_agg<false>(a, v4, (hipStream_t)stream);
}

size_t ctgcn_hub_workspace_bytes(int32_t n_long, int32_t hub_split, int32_t slots, int32_t d)
{
    return hub_workspace_bytes_(n_long, hub_split, slots, d);
}

int32_t ctgcn_hub_split_entries(void) { return HUB_SPLIT_ENTRIES; }

size_t ctgcn_workspace_bytes(int op, int64_t n, int64_t nnz, int32_t d, int32_t K)
{
    (void)nnz; (void)d; (void)K;
    if (op == CTGCN_OP_KCORE) {
        const size_t nn = (size_t)(n > 0 ? n : 0);
        return align_up(sizeof(KcoreCtl), 256) + align_up(nn * 4, 256) /*deg*/ + align_up((nn + 31) / 32 * 4, 256) /*claimed*/
               + 2 * align_up(nn * 4, 256) /*pool*/;
    }
    if (op == CTGCN_OP_INGEST) return ctgcn_ingest_workspace_bytes_(n, nnz);   /* nnz = number of edge rows m */
    return 0;
}

int ctgcn_kcore_i32(int64_t n, const int32_t *row_ptr, const int32_t *col_idx, int32_t *core,
                    void *workspace, size_t workspace_bytes, int32_t level_cap, int32_t *max_core_host, void *stream)
{
    if (n < 0 || n > 0x7fffffffLL) return fail(CTGCN_E_INVALID, "kcore: n=%lld out of range", (long long)n);
    if (max_core_host) *max_core_host = 0;
    if (n == 0) return CTGCN_OK;
    if (!row_ptr || !core || !workspace) return fail(CTGCN_E_INVALID, "kcore: null pointer");
    const size_t need = ctgcn_workspace_bytes(CTGCN_OP_KCORE, n, 0, 0, 0);
    if (workspace_bytes < need) return fail(CTGCN_E_WORKSPACE, "kcore: workspace %zu < %zu bytes", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    KcoreCtl *ctl = (KcoreCtl *)ws; ws += align_up(sizeof(KcoreCtl), 256);
    int32_t *deg = (int32_t *)ws; ws += align_up((size_t)n * 4, 256);
    unsigned *claimed = (unsigned *)ws; const size_t claimed_bytes = align_up(((size_t)n + 31) / 32 * 4, 256); ws += claimed_bytes;
    int32_t *pool_v = (int32_t *)ws; ws += align_up((size_t)n * 4, 256);
    int32_t *pool_prev = (int32_t *)ws;

    const int nn = (int)n;
    // Two algorithms, the same integers.  The level-synchronous peel costs one dependent cascade per level (~86 us each on the config-5
    // snapshots): unbeatable when the loader's max_core caps it at a few levels (0.58 ms at cap 8 against 0.9 ms of sweeps), 7.3 ms for all 84
    // levels.  h-index sweeps cost ~40 sweeps whatever the depth: 4.0 ms there.  Default: peel up to a cap of 16 levels, sweeps beyond and for
    // the exact core numbers; CTGCN_KCORE=peel / hindex forces one (tools/kcore_bench.py, profiles/r05_kcore_hindex.txt).
    static const int forced = [] { const char *e = getenv("CTGCN_KCORE"); return !e ? 0 : (!strcmp(e, "peel") ? 1 : (!strcmp(e, "hindex") ? 2 : 0)); }();
    const bool use_peel = forced == 1 || (forced == 0 && level_cap > 0 && level_cap <= 16);
    if (!use_peel) {
        KhCtl *kc = (KhCtl *)ctl;                         // 256 bytes at the head of the workspace
        uint8_t *flags = (uint8_t *)pool_v;               // 2 n bytes of the peel's 8 n byte pool
        int32_t *hub_list = pool_prev;                    // up to n hub vertices
        const int capv = level_cap > 0 ? level_cap : 0x7fffffff;
        HIP_TRY(hipMemsetAsync(kc, 0, sizeof(KhCtl), st));
        hipLaunchKernelGGL(kcore_hindex_init_kernel, dim3((unsigned)(((int64_t)nn * 8 + 255) / 256)), dim3(256), 0, st, nn, capv, row_ptr, col_idx, deg, flags);
        const unsigned blocks = (unsigned)((nn + KH_CHUNK - 1) / KH_CHUNK);
        static const int FULL = [] { const char *e = getenv("CTGCN_KCORE_FULL"); const int v = e ? atoi(e) : 3; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
        // FULL sweeps recompute every vertex (the last of them sets flags).  No host check between sweeps (round 5 read the counters back every 8
        // sweeps: five or six stream synchronisations inside one call): a sweep that finds its predecessor's counter at zero sets ctl->done and
        // every kernel queued behind it returns at once (~4 us each), so the sweeps are queued in ONE batch sized for power-law graphs (48; the
        // config-5 snapshots need 36 - 40), the finish kernel behind them, and the control block is read once.  Graphs that need more (long
        // paths: one sweep per hop) continue in batches of 32 with a read each.
        KhCtl hk{};
        int sweep = 0;
        static const bool trace = getenv("CTGCN_KCORE_TRACE") != nullptr;       // diagnostic: one sweep per batch, the flags every sweep set on stderr
        for (int batch = trace ? 1 : 48;; batch = trace ? 1 : 32) {
            for (int b = 0; b < batch; ++b, ++sweep) {
                const int mark = sweep >= FULL - 1 ? 1 : 0, check = sweep >= FULL ? 1 : 0;
                hipLaunchKernelGGL(kcore_hindex_kernel, dim3(blocks), dim3(256), 0, st, nn, capv, sweep < FULL ? 1 : 0, mark, check, sweep,
                                   row_ptr, col_idx, deg, flags + (size_t)(sweep & 1) * nn, flags + (size_t)((sweep + 1) & 1) * nn, kc, hub_list);
                // 1 024 blocks (33 KB of LDS each: four per CU): the ~850 hubs of a config-5 snapshot that move in nearly every sweep (they converge
                // last) are recomputed in ONE round of blocks instead of four — 30 - 46 us per sweep with 256 blocks, x 37 sweeps = 1.4 of 3.7 ms
                hipLaunchKernelGGL(kcore_hindex_hub_kernel, dim3(1024), dim3(256), 0, st, mark, check, sweep, row_ptr, col_idx, deg,
                                   flags + (size_t)((sweep + 1) & 1) * nn, kc, hub_list);
            }
            // (the finish kernel's maximum is an atomicMax: a batch that ended before the fixed point has left the maximum of its unfinished
            // upper bounds there — found by the 300 000-vertex path of tests/test_gpu_kernels.py, 150 000 sweeps: max core 2 instead of 1)
            HIP_TRY(hipMemsetAsync(&kc->max_core, 0, sizeof(int), st));
            hipLaunchKernelGGL(kcore_hindex_finish_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, nn, deg, core, kc);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&hk, kc, sizeof(KhCtl), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (trace) fprintf(stderr, "kcore: sweep %d set %d flags (%d hubs queued), done %d\n", sweep - 1, hk.marks[(sweep - 1) & 15], hk.hubs[(sweep - 1) & 15], hk.done);
            if (hk.done || (sweep - 1 >= FULL - 1 && hk.marks[(sweep - 1) & 15] == 0)) break;      // (the batch's last sweep may be the marking sweep that set no flag)
            if (sweep > nn + 64) return fail(CTGCN_E_HIP, "kcore: h-index sweeps did not converge");
        }
        if (max_core_host) *max_core_host = hk.max_core;
        return CTGCN_OK;
    }
    HIP_TRY(hipMemsetAsync(ctl, 0, sizeof(KcoreCtl), st));
    HIP_TRY(hipMemsetAsync(claimed, 0, claimed_bytes, st));
    hipLaunchKernelGGL(kcore_init_kernel, dim3((unsigned)(((int64_t)nn * 8 + 255) / 256)), dim3(256), 0, st, nn, row_ptr, col_idx, deg);
    HIP_TRY(hipGetLastError());

    int blocks = (nn + 1023) / 1024;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    const int chunk = (nn + blocks - 1) / blocks;
    constexpr int BATCH = 16;
    const int cap = level_cap > 0 ? level_cap : 0x7fffffff;      // peel levels 0 .. cap-1 only
    KcoreCtl h{};
    for (int level = 0; level < cap; level += BATCH) {
        for (int b = 0; b < BATCH && level + b < cap; ++b)
            hipLaunchKernelGGL(kcore_level_kernel, dim3(blocks), dim3(256), 0, st, nn, level + b, chunk, row_ptr, col_idx, deg,
                               claimed, pool_v, pool_prev, ctl);
        // the result of what has been peeled so far, then ONE read: with the loader's cap (<= 16 levels) the call synchronises once
        hipLaunchKernelGGL(kcore_copy_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, nn, cap, deg, core);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&h, ctl, sizeof(KcoreCtl), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h.visited >= nn) break;
        if (level > nn) return fail(CTGCN_E_HIP, "kcore: did not converge (visited %d of %d)", h.visited, nn);
    }
    if (max_core_host) *max_core_host = h.visited >= nn ? h.max_core : cap;     // survivors exist: max core >= cap, reported as cap
    return CTGCN_OK;
}

int ctgcn_edge_levels_i32(int64_t n, const int32_t *row_ptr, const int32_t *col_idx,
                          const float *val, const int32_t *core, int32_t *level,
                          int64_t *count, double *wsum, int32_t hist_len, void *stream)
{
    if (n < 0 || hist_len < 0) return fail(CTGCN_E_INVALID, "edge_levels: bad sizes");
    if (n == 0) return CTGCN_OK;
    if (!row_ptr || !core) return fail(CTGCN_E_INVALID, "edge_levels: null pointer");
    if (wsum && !val) return fail(CTGCN_E_INVALID, "edge_levels: val required for wsum");
    int64_t blocks = (n + 15) / 16;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(edge_level_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, row_ptr, col_idx, val, core,
                       level, (unsigned long long *)count, wsum, (int)hist_len);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_slot_reorder(int64_t n, int32_t K, const int32_t *row_ptr, const int32_t *col_idx,
                       const float *val, const int32_t *level, const uint8_t *slot_of_level,
                       int32_t table_len, int32_t *col_out, float *val_out, uint8_t *slot_out, void *stream)
{
    if (n < 0 || K < 1 || K > CTGCN_MAX_SLOTS || table_len < 1) return fail(CTGCN_E_INVALID, "slot_reorder: bad sizes");
    if (n == 0) return CTGCN_OK;
    if (!row_ptr || !slot_of_level) return fail(CTGCN_E_INVALID, "slot_reorder: null pointer");
    if (col_idx && (col_out == col_idx || val_out == val)) return fail(CTGCN_E_INVALID, "slot_reorder: outputs alias inputs");
    const int64_t blocks = (n + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(CTGCN_E_UNSUPPORTED, "slot_reorder: grid too large");
    hipLaunchKernelGGL(slot_reorder_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, (int)K, row_ptr, col_idx, val,
                       level, slot_of_level, (int)table_len, col_out, val_out, slot_out);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int32_t ctgcn_compute_units(void)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    return (int32_t)persistent_cus(cus);
}

int64_t ctgcn_gru_row_granule(void)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        cus = 256;
    return (int64_t)GRU_BM * cus;
}

int ctgcn_gru_seq_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gi, const float *w_hh,
                      const float *b_hn, const float *ln_weight, const float *ln_bias, float ln_eps,
                      int reduce_sum, float *out, int64_t ld_out, float *gates_out, int split_bf16, int gi_blocked,
                      const int32_t *row_order, const uint32_t *tile_mask, const int32_t *tile_base, void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_seq: hidden=%d, only %d is built", hidden, GRU_H);
    if ((row_order == nullptr) != (tile_mask == nullptr) || (row_order == nullptr) != (tile_base == nullptr))
        return fail(CTGCN_E_INVALID, "gru_seq: row_order, tile_mask and tile_base come together");
    if (row_order && (!reduce_sum || gates_out || gi_blocked || split_bf16 != CTGCN_SPLIT_F16X2 || steps > 64))
        return fail(CTGCN_E_UNSUPPORTED, "gru_seq: a row plan needs the sum-over-steps form, CTGCN_SPLIT_F16X2, the plain gi layout and steps <= 64");
    if (rows < 0 || steps < 1) return fail(CTGCN_E_INVALID, "gru_seq: bad sizes rows=%lld steps=%d", (long long)rows, steps);
    if (rows == 0) return CTGCN_OK;
    if (!gi || !w_hh || !out) return fail(CTGCN_E_INVALID, "gru_seq: null pointer");
    if (!aligned16(w_hh) || (reinterpret_cast<uintptr_t>(out) & 7u) || (ln_weight && (reinterpret_cast<uintptr_t>(ln_weight) & 7u)))
        return fail(CTGCN_E_INVALID, "gru_seq: w_hh must be 16-byte aligned, out / ln_weight 8-byte aligned");
    GruArgs a{};
    a.rows = rows; a.steps = steps; a.gi = gi; a.whh = w_hh; a.bhn = b_hn; a.gamma = ln_weight; a.beta = ln_bias;
    a.eps = ln_eps; a.reduce_sum = reduce_sum ? 1 : 0; a.out = out; a.gates = gates_out; a.gi_blocked = gi_blocked ? 1 : 0;
    a.ldo = ld_out > 0 ? ld_out : GRU_H;
    a.order = row_order; a.tmask = tile_mask; a.tbase = tile_base;
    if (ld_out > 0 && (!reduce_sum || ld_out < GRU_H || (ld_out & 1))) return fail(CTGCN_E_INVALID, "gru_seq: ld_out=%lld needs reduce_sum and an even value >= %d", (long long)ld_out, GRU_H);
    if (gi_blocked && split_bf16 != CTGCN_SPLIT_F16X2) return fail(CTGCN_E_INVALID, "gru_seq: the blocked gi layout belongs to CTGCN_SPLIT_F16X2");
    if (gates_out && (reduce_sum || ln_weight)) return fail(CTGCN_E_INVALID, "gru_seq: gates_out needs reduce_sum == 0 and no LayerNorm (raw h sequence)");
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    const int64_t ntiles = (rows + GRU_BM - 1) / GRU_BM;
    const int64_t blocks = ntiles < cus ? ntiles : cus;          // persistent: one 8-wave block per CU
    if (split_bf16 == 2 && a.reduce_sum && row_order && steps > 32)      // two mask words per tile
        hipLaunchKernelGGL((gru_seq_h2_kernel<true, false, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (split_bf16 == 2 && a.reduce_sum)
        hipLaunchKernelGGL((gru_seq_h2_kernel<true, false>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (split_bf16 == 2 && a.gates)
        hipLaunchKernelGGL((gru_seq_h2_kernel<false, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (split_bf16 == 2)
        hipLaunchKernelGGL((gru_seq_h2_kernel<false, false>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (a.reduce_sum && split_bf16)
        hipLaunchKernelGGL((gru_seq_x3_kernel<true, false>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (split_bf16 && a.gates)
        hipLaunchKernelGGL((gru_seq_x3_kernel<false, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (split_bf16)
        hipLaunchKernelGGL((gru_seq_x3_kernel<false, false>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (a.reduce_sum)
        hipLaunchKernelGGL((gru_seq_kernel<true, false>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (a.gates)
        hipLaunchKernelGGL((gru_seq_kernel<false, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((gru_seq_kernel<false, false>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_layernorm_bwd_f32(int64_t rows, int32_t steps, int32_t hidden, const float *h, const float *dy, int64_t ld_dy, const float *gamma, float eps,
                            float *dx, float *partial, int32_t n_partial, const int32_t *dy_rows, void *stream)
{
    if (ld_dy == 0) ld_dy = GRU_H;
    if (ld_dy < GRU_H || (ld_dy & 1)) return fail(CTGCN_E_INVALID, "layernorm_bwd: ld_dy=%lld must be even and >= %d", (long long)ld_dy, GRU_H);
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "layernorm_bwd: hidden=%d, only %d is built", hidden, GRU_H);
    if (rows < 0 || steps < 1 || n_partial < 1 || n_partial > 65535) return fail(CTGCN_E_INVALID, "layernorm_bwd: bad sizes rows=%lld steps=%d n_partial=%d", (long long)rows, steps, n_partial);
    if (!partial) return fail(CTGCN_E_INVALID, "layernorm_bwd: null pointer");
    if (rows == 0) { HIP_TRY(hipMemsetAsync(partial, 0, (size_t)n_partial * 2 * GRU_H * sizeof(float), (hipStream_t)stream)); return CTGCN_OK; }
    if (!h || !dy || !gamma || !dx) return fail(CTGCN_E_INVALID, "layernorm_bwd: null pointer");
    if ((reinterpret_cast<uintptr_t>(h) & 7u) || (reinterpret_cast<uintptr_t>(dy) & 7u) || (reinterpret_cast<uintptr_t>(dx) & 7u) || (reinterpret_cast<uintptr_t>(gamma) & 7u))
        return fail(CTGCN_E_INVALID, "layernorm_bwd: h / dy / dx / gamma must be 8-byte aligned");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)n_partial), dim3(256), 0, (hipStream_t)stream, rows, steps, h, dy, ld_dy, gamma, eps, dx, partial, dy_rows);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_layer_f32(int64_t rows, int32_t steps, int32_t d_in, int32_t hidden, const float *x, int64_t ldx, const float *w_ih,
                        const float *w_hh, const float *bias_gi, const float *b_hn, const float *ln_weight, const float *ln_bias,
                        float ln_eps, int reduce_sum, float *out, int64_t ld_out, float *gates_out,
                        const int64_t *step_offsets, int64_t ld_row, void *stream)
{
    if (step_offsets && (ld_row < d_in || (ld_row & 3))) return fail(CTGCN_E_INVALID, "gru_layer: ld_row=%lld must be a multiple of 4 and >= d_in", (long long)ld_row);
    if (hidden != GRU_H || d_in != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_layer: only d_in = hidden = %d is built (got %d, %d)", GRU_H, d_in, hidden);
    if (gates_out && (reduce_sum || ln_weight)) return fail(CTGCN_E_INVALID, "gru_layer: gates_out needs reduce_sum == 0 and no LayerNorm (raw h sequence)");
    if (gates_out && !aligned16(gates_out)) return fail(CTGCN_E_INVALID, "gru_layer: gates_out must be 16-byte aligned");
    if (rows < 0 || steps < 1 || ldx < d_in) return fail(CTGCN_E_INVALID, "gru_layer: bad sizes rows=%lld steps=%d ldx=%lld", (long long)rows, steps, (long long)ldx);
    if (rows == 0) return CTGCN_OK;
    if (!x || !w_ih || !w_hh || !out) return fail(CTGCN_E_INVALID, "gru_layer: null pointer");
    if (!aligned16(x) || !aligned16(w_ih) || !aligned16(w_hh) || (ldx % 4) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
        (ln_weight && (reinterpret_cast<uintptr_t>(ln_weight) & 7u)) || (ln_bias && (reinterpret_cast<uintptr_t>(ln_bias) & 7u)))
        return fail(CTGCN_E_INVALID, "gru_layer: x / weights must be 16-byte aligned, ldx a multiple of 4, out / LayerNorm vectors 8-byte aligned");
    const int64_t ldo = ld_out > 0 ? ld_out : GRU_H;
    if (ld_out > 0 && (!reduce_sum || ld_out < GRU_H || (ld_out & 1)))
        return fail(CTGCN_E_INVALID, "gru_layer: ld_out=%lld needs reduce_sum and an even value >= %d", (long long)ld_out, GRU_H);
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    const int64_t ntiles = (rows + LY_BM - 1) / LY_BM;
    const int64_t blocks = ntiles < cus ? ntiles : cus;          // persistent: one 4-wave block per CU (one wave per SIMD, 512 registers)
    LayerArgs a{};
    a.rows = rows; a.steps = steps; a.x = x; a.ldx = ldx; a.wih = w_ih; a.whh = w_hh; a.bias_gi = bias_gi; a.bhn = b_hn;
    a.gamma = ln_weight; a.beta = ln_bias; a.eps = ln_eps; a.out = out; a.ldo = ldo; a.gates = gates_out;
    a.step_off = step_offsets; a.ld_row = ld_row;
    static const int nw_ = [] { const char *e = getenv("CTGCN_GRU_LAYER_WAVES"); return e && atoi(e) == 4 ? 4 : 8; }();
    if (step_offsets && nw_ != 8) return fail(CTGCN_E_UNSUPPORTED, "gru_layer: step_offsets belong to the 8-wave kernels");
    if (gates_out) {
        // training's recompute pass: raw h sequence + gates straight from the layer kernel (the kernel pair wrote and re-read gi for this)
        const int64_t nt8 = (rows + 15) / 16;
#ifdef CTGCN_LAYER_TIMELINE
        a.timeline = nullptr;
#endif
        hipLaunchKernelGGL((gru_layer8_h2_kernel<false, false, true>), dim3((unsigned)(nt8 < cus ? nt8 : cus)), dim3(512), 0, (hipStream_t)stream, a);
        HIP_TRY(hipGetLastError());
        return CTGCN_OK;
    }
    // sum-over-steps form: the 8-wave kernel (4.6 ms per 1M x 8 call; 5.2 for the 4-wave one, 6.4 for the kernel pair); CTGCN_GRU_LAYER_WAVES=4 forces the latter
    static const int nw = [] { const char *e = getenv("CTGCN_GRU_LAYER_WAVES"); return e && atoi(e) == 4 ? 4 : 8; }();
    if (nw == 8 && reduce_sum) {
        const int64_t nt8 = (rows + 15) / 16;
#ifdef CTGCN_LAYER_TIMELINE
        const unsigned nb8 = (unsigned)(nt8 < cus ? nt8 : cus);
        const char *tl_file = timeline_begin(a, nb8, stream);
#endif
        hipLaunchKernelGGL((gru_layer8_h2_kernel<false, true>), dim3((unsigned)(nt8 < cus ? nt8 : cus)), dim3(512), 0, (hipStream_t)stream, a);
#ifdef CTGCN_LAYER_TIMELINE
        timeline_end(a, nb8, tl_file, stream);
#endif
    } else if (nw == 8) {
        // per-step form (temporal GRU): LayerNorm(h_t) of every unit leaves through an fp32 staging buffer in LDS (free since the
        // weights take 96 instead of 120 KB), two rows per wave, during the next unit
        const int64_t nt8 = (rows + 15) / 16;
#ifdef CTGCN_LAYER_TIMELINE
        a.timeline = nullptr;
#endif
        hipLaunchKernelGGL((gru_layer8_h2_kernel<false, false>), dim3((unsigned)(nt8 < cus ? nt8 : cus)), dim3(512), 0, (hipStream_t)stream, a);
    } else {
        if (reduce_sum) hipLaunchKernelGGL((gru_layer_h2_kernel<true, 4>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gru_layer_h2_kernel<false, 4>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_layer_presplit_f32(int64_t rows, int32_t steps, int32_t hidden, const void *planes, const float *w_ih, const float *w_hh,
                                 const float *bias_gi, const float *b_hn, const float *ln_weight, const float *ln_bias, float ln_eps,
                                 float *out, int64_t ld_out, const int32_t *row_order, const uint32_t *tile_mask, void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_layer_presplit: only d_in = hidden = %d is built (got %d)", GRU_H, hidden);
    if ((row_order == nullptr) != (tile_mask == nullptr)) return fail(CTGCN_E_INVALID, "gru_layer_presplit: row_order and tile_mask come together");
    if (row_order && steps > 64) return fail(CTGCN_E_UNSUPPORTED, "gru_layer_presplit: a row plan needs steps <= 64");
    if (rows < 0 || steps < 1) return fail(CTGCN_E_INVALID, "gru_layer_presplit: bad sizes rows=%lld steps=%d", (long long)rows, steps);
    if (rows == 0) return CTGCN_OK;
    if (!planes || !w_ih || !w_hh || !out) return fail(CTGCN_E_INVALID, "gru_layer_presplit: null pointer");
    if ((reinterpret_cast<uintptr_t>(planes) & 255u) || !aligned16(w_ih) || !aligned16(w_hh) || (reinterpret_cast<uintptr_t>(out) & 7u) ||
        (ln_weight && (reinterpret_cast<uintptr_t>(ln_weight) & 7u)) || (ln_bias && (reinterpret_cast<uintptr_t>(ln_bias) & 7u)))
        return fail(CTGCN_E_INVALID, "gru_layer_presplit: planes 256-byte, weights 16-byte, out / LayerNorm vectors 8-byte aligned");
    const int64_t ldo = ld_out > 0 ? ld_out : GRU_H;
    if (ld_out > 0 && (ld_out < GRU_H || (ld_out & 1))) return fail(CTGCN_E_INVALID, "gru_layer_presplit: ld_out must be even and >= %d", GRU_H);
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    LayerArgs a{};
    a.rows = rows; a.steps = steps; a.x = nullptr; a.ldx = GRU_H; a.wih = w_ih; a.whh = w_hh; a.bias_gi = bias_gi; a.bhn = b_hn;
    a.gamma = ln_weight; a.beta = ln_bias; a.eps = ln_eps; a.out = out; a.ldo = ldo;
    // the layout ctgcn_core_aggregate_split_f32 writes for d = 128 (kp = 128): plane 1, plane 2, row scales
    const size_t nrow = (size_t)rows * steps;
    a.xp1 = (const _Float16 *)planes;
    a.xp2 = a.xp1 + nrow * GRU_H;
    a.xps = (const float *)(a.xp2 + nrow * GRU_H);
    a.order = row_order; a.tmask = tile_mask;
    const int64_t nt8 = (rows + 15) / 16;
#ifdef CTGCN_LAYER_TIMELINE
    const unsigned nb8 = (unsigned)(nt8 < cus ? nt8 : cus);
    const char *tl_file = timeline_begin(a, nb8, stream);
#endif
    if (row_order && steps > 32)       // two mask words per tile (tile_mask[2 tile], [2 tile + 1])
        hipLaunchKernelGGL((gru_layer8_h2_kernel<true, true, false, true>), dim3((unsigned)(nt8 < cus ? nt8 : cus)), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((gru_layer8_h2_kernel<true, true>), dim3((unsigned)(nt8 < cus ? nt8 : cus)), dim3(512), 0, (hipStream_t)stream, a);
#ifdef CTGCN_LAYER_TIMELINE
    timeline_end(a, nb8, tl_file, stream);
#endif
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_layer_presplit_save_f32(int64_t rows, int32_t steps, int32_t hidden, const void *planes, int64_t plane_rows, int64_t first_row,
                                      const float *w_ih, const float *w_hh, const float *bias_gi, const float *b_hn, const uint32_t *tile_mask,
                                      float *gates_out, float *hseq_out, float *presum_out, void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_layer_presplit_save: only d_in = hidden = %d is built (got %d)", GRU_H, hidden);
    if (rows < 0 || steps < 1 || first_row < 0 || (first_row & 15) || (first_row + rows) * steps > plane_rows)
        return fail(CTGCN_E_INVALID, "gru_layer_presplit_save: bad sizes rows=%lld steps=%d first_row=%lld (a multiple of 16) plane_rows=%lld",
                    (long long)rows, steps, (long long)first_row, (long long)plane_rows);
    if (tile_mask && steps > 64) return fail(CTGCN_E_UNSUPPORTED, "gru_layer_presplit_save: a row plan needs steps <= 64");
    if (rows == 0) return CTGCN_OK;
    if (!planes || !w_ih || !w_hh || !gates_out || !hseq_out || !presum_out) return fail(CTGCN_E_INVALID, "gru_layer_presplit_save: null pointer");
    if ((reinterpret_cast<uintptr_t>(planes) & 255u) || !aligned16(w_ih) || !aligned16(w_hh) || !aligned16(gates_out) || !aligned16(hseq_out) || !aligned16(presum_out))
        return fail(CTGCN_E_INVALID, "gru_layer_presplit_save: planes 256-byte, everything else 16-byte aligned");
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    LayerArgs a{};
    a.rows = rows; a.steps = steps; a.x = nullptr; a.ldx = GRU_H; a.wih = w_ih; a.whh = w_hh; a.bias_gi = bias_gi; a.bhn = b_hn;
    a.gamma = nullptr; a.beta = nullptr; a.eps = 0.f; a.out = nullptr; a.ldo = GRU_H;
    // the layout ctgcn_core_aggregate_split_f32 writes for d = 128: plane 1, plane 2, row scales, plane_rows rows each; this call takes the
    // sequences [first_row, first_row + rows) of it (tile_mask already points at first_row / 16)
    a.xp1 = (const _Float16 *)planes + (size_t)first_row * steps * GRU_H;
    a.xp2 = (const _Float16 *)planes + ((size_t)plane_rows + (size_t)first_row * steps) * GRU_H;
    a.xps = (const float *)((const _Float16 *)planes + 2 * (size_t)plane_rows * GRU_H) + (size_t)first_row * steps;
    a.order = nullptr; a.tmask = tile_mask;
    a.gates = gates_out; a.hseq = hseq_out; a.presum = presum_out; a.gates3 = 1;
#ifdef CTGCN_LAYER_TIMELINE
    a.timeline = nullptr;
#endif
    const int64_t nt8 = (rows + 15) / 16;
    if (tile_mask && steps > 32)       // two mask words per tile (round 6: training under the row plan for core lists of 33-64 matrices)
        hipLaunchKernelGGL((gru_layer8_h2_kernel<true, true, true, true>), dim3((unsigned)(nt8 < cus ? nt8 : cus)), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((gru_layer8_h2_kernel<true, true, true>), dim3((unsigned)(nt8 < cus ? nt8 : cus)), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

size_t ctgcn_group_table_bytes(int32_t groups)
{
    if (groups < 1) return 0;
    size_t per = sizeof(AggSplitGroup) > sizeof(LayerArgs) ? sizeof(AggSplitGroup) : sizeof(LayerArgs);
    if (sizeof(GruArgs) > per) per = sizeof(GruArgs);
    return ((size_t)groups * per + 255) / 256 * 256 + 3 * sizeof(int32_t) * 1024;      // + the block map of the persistent launches (<= 1024 blocks)
}

int ctgcn_core_aggregate_split_group_f32(int32_t groups, int64_t n_rows, int32_t d, const ctgcn_agg_split_group_t *g, void *table, size_t table_bytes,
                                         void *shadow, void *stream)
{
    if (groups < 1 || groups > 1024 || !g) return fail(CTGCN_E_INVALID, "core_aggregate_split_group: groups=%d", groups);
    if ((d & 3) || d < 32 || d > 512) return fail(CTGCN_E_UNSUPPORTED, "core_aggregate_split_group: needs d %% 4 == 0, 32 <= d <= 512 (got %d)", d);
    if (n_rows < 1) return fail(CTGCN_E_INVALID, "core_aggregate_split_group: n_rows=%lld", (long long)n_rows);
    if (!table || (reinterpret_cast<uintptr_t>(table) & 255u) || table_bytes < ctgcn_group_table_bytes(groups))
        return fail(CTGCN_E_WORKSPACE, "core_aggregate_split_group: table must be 256-byte aligned and hold ctgcn_group_table_bytes(groups) bytes");
    std::vector<AggSplitGroup> host((size_t)groups);
    const AggPlan p = plan_for(d, true);
    const int32_t kp = (d + 63) / 64 * 64;
    const bool layer_form = d == GRU_H && !g[0].planes1;      // consumer = the GRU layer kernel: planes with holes in the group's own workspace
    for (int i = 0; i < groups; ++i) {
        const ctgcn_agg_split_group_t &q = g[i];
        if (q.K < 1 || q.K > CTGCN_MAX_SLOTS || !q.row_ptr || !q.X || (!q.slot && q.K != 1) || q.ldx < d || (q.ldx & 3) || !aligned16(q.X))
            return fail(CTGCN_E_INVALID, "core_aggregate_split_group: bad arguments of group %d", i);
        if ((q.row_order == nullptr) != (q.tile_mask == nullptr) || (q.row_order && q.K > 32))
            return fail(CTGCN_E_INVALID, "core_aggregate_split_group: group %d: row_order and tile_mask come together, K <= 32 under a plan", i);
        AggSplitGroup &h = host[i];
        std::memset((void *)&h, 0, sizeof h);           // padding included: the table is compared byte-wise with its shadow
        AggArgs &a = h.a;
        if (layer_form) {
            if (!q.workspace || (reinterpret_cast<uintptr_t>(q.workspace) & 255u) || q.planes1 || q.tile_base)
                return fail(CTGCN_E_INVALID, "core_aggregate_split_group: group %d: the layer-kernel form takes a workspace per group and no tile_base", i);
            if (q.workspace_bytes < ctgcn_core_aggregate_split_workspace_bytes(n_rows, d, q.K, 1, 0))
                return fail(CTGCN_E_WORKSPACE, "core_aggregate_split_group: workspace of group %d too small", i);
            const int64_t rows = n_rows * q.K;
            h.p1 = (_Float16 *)q.workspace; h.p2 = h.p1 + (size_t)rows * GRU_H; h.scale = (float *)(h.p2 + (size_t)rows * GRU_H);
            a.tbase = nullptr; a.tile_shift = 4;
        } else {
            // consumer = the split GEMM: the group's operand rows sit at [first_row, first_row + operand_rows) of planes SHARED by the window
            // (planes1 / planes2 / scales point at the group's first row) so that ONE grouped GEMM launch walks all of them; a row plan makes
            // them compact (tile 64: tile_base), without one every (node, core) row is written
            if (!q.planes1 || !q.planes2 || !q.scales || (reinterpret_cast<uintptr_t>(q.planes1) & 15u) || (reinterpret_cast<uintptr_t>(q.planes2) & 15u))
                return fail(CTGCN_E_INVALID, "core_aggregate_split_group: group %d: the GEMM form takes planes1 / planes2 / scales", i);
            if ((q.row_order != nullptr) != (q.tile_base != nullptr))
                return fail(CTGCN_E_INVALID, "core_aggregate_split_group: group %d: the GEMM form's row plan comes with tile_base", i);
            h.p1 = (_Float16 *)q.planes1; h.p2 = (_Float16 *)q.planes2; h.scale = q.scales;
            a.tbase = q.tile_base; a.tile_shift = q.tile_base ? 6 : 4;
        }
        a.n = n_rows; a.d = d; a.K = q.K; a.row_ptr = q.row_ptr; a.col = q.col_idx; a.val = q.val; a.slot = q.slot;
        a.src = q.X; a.ldsrc = q.ldx; a.out = nullptr; a.out_ld = (int64_t)q.K * d; a.flags = q.flags;
        a.n_long = 0; a.long_thresh = 0x7fffffff;          // hub rows are the caller's business (none in the windows this serves)
        a.order = q.row_order; a.tmask = q.tile_mask;
        a.chunks = p.chunks; a.passes = p.passes; a.hub_split = 1;
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctgcn_table::upload(table, host.data(), host.size() * sizeof(AggSplitGroup), shadow, st));
    const int rows_per_block = p.chunks <= 32 ? 8 : 4;
    const int64_t bpg = (n_rows + rows_per_block - 1) / rows_per_block;
    if (bpg * groups > 0x7fffffffLL) return fail(CTGCN_E_UNSUPPORTED, "core_aggregate_split_group: grid too large");
    const dim3 grid((unsigned)(bpg * groups));
    if (p.chunks <= 32) hipLaunchKernelGGL(agg_fwd_split32_group_kernel, grid, dim3(256), 0, st, (const AggSplitGroup *)table, (int32_t)bpg, kp, 1.f);
    else if (p.chunks <= 64) hipLaunchKernelGGL(agg_fwd_split_group_kernel<1>, grid, dim3(256), 0, st, (const AggSplitGroup *)table, (int32_t)bpg, kp, 1.f);
    else hipLaunchKernelGGL(agg_fwd_split_group_kernel<2>, grid, dim3(256), 0, st, (const AggSplitGroup *)table, (int32_t)bpg, kp, 1.f);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_transpose_bias_group_f32(int32_t groups, int64_t n, int32_t d, const float *const *w, int64_t ldw, const float *const *bias, float *const *out,
                                   int64_t ldo, void *table, size_t table_bytes, void *shadow, void *stream)
{
    if (groups < 1 || groups > 1024 || n < 1 || d < 1 || ldw < n || ldo < d || !w || !out) return fail(CTGCN_E_INVALID, "transpose_bias_group: bad arguments");
    if (!table || (reinterpret_cast<uintptr_t>(table) & 255u) || table_bytes < ctgcn_group_table_bytes(groups))
        return fail(CTGCN_E_WORKSPACE, "transpose_bias_group: table must be 256-byte aligned and hold ctgcn_group_table_bytes(groups) bytes");
    std::vector<TransposeGroup> host((size_t)groups);
    bool vec = !(ldw & 3) && !(ldo & 3);
    for (int i = 0; i < groups; ++i) {
        if (!w[i] || !out[i]) return fail(CTGCN_E_INVALID, "transpose_bias_group: null pointer in group %d", i);
        host[i] = TransposeGroup{w[i], bias ? bias[i] : nullptr, out[i]};
        vec = vec && aligned16(w[i]) && aligned16(out[i]);
    }
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(ctgcn_table::upload(table, host.data(), host.size() * sizeof(TransposeGroup), shadow, st));
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((d + 63) / 64), (unsigned)groups);
    if (vec) hipLaunchKernelGGL(transpose_bias_group_kernel<true>, grid, dim3(256), 0, st, n, d, (const TransposeGroup *)table, ldw, ldo);
    else hipLaunchKernelGGL(transpose_bias_group_kernel<false>, grid, dim3(256), 0, st, n, d, (const TransposeGroup *)table, ldw, ldo);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_seq_group_f32(int32_t groups, int64_t rows, int32_t hidden, const ctgcn_gru_seq_group_t *g, void *table, size_t table_bytes, void *shadow,
                            void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_seq_group: only hidden = %d is built (got %d)", GRU_H, hidden);
    if (groups < 1 || groups > 1024 || !g || rows < 1) return fail(CTGCN_E_INVALID, "gru_seq_group: groups=%d rows=%lld", groups, (long long)rows);
    if (!table || (reinterpret_cast<uintptr_t>(table) & 255u) || table_bytes < ctgcn_group_table_bytes(groups))
        return fail(CTGCN_E_WORKSPACE, "gru_seq_group: table must be 256-byte aligned and hold ctgcn_group_table_bytes(groups) bytes");
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    if (cus > 1024) cus = 1024;
    if (groups > cus) return fail(CTGCN_E_UNSUPPORTED, "gru_seq_group: more snapshots (%d) than CUs (%d): split the window", groups, cus);
    std::vector<GruArgs> host((size_t)groups);
    const int64_t ntiles = (rows + GRU_BM - 1) / GRU_BM;
    double total = 0.0;
    for (int i = 0; i < groups; ++i) {
        const ctgcn_gru_seq_group_t &q = g[i];
        if (q.steps < 1 || !q.gi || !q.w_hh || !q.out || !aligned16(q.gi) || !aligned16(q.w_hh) || (reinterpret_cast<uintptr_t>(q.out) & 7u) ||
            (q.ld_out > 0 && (q.ld_out < GRU_H || (q.ld_out & 1))) || (q.row_order == nullptr) != (q.tile_mask == nullptr) ||
            (q.row_order == nullptr) != (q.tile_base == nullptr) || (q.row_order && q.steps > 32))
            return fail(CTGCN_E_INVALID, "gru_seq_group: bad arguments of group %d", i);
        GruArgs &a = host[i];
        std::memset((void *)&a, 0, sizeof a);
        a.rows = rows; a.steps = q.steps; a.gi = q.gi; a.whh = q.w_hh; a.bhn = q.b_hn; a.gamma = q.ln_weight; a.beta = q.ln_bias; a.eps = q.ln_eps;
        a.reduce_sum = 1; a.out = q.out; a.gates = nullptr; a.gi_blocked = 0; a.ldo = q.ld_out > 0 ? q.ld_out : GRU_H;
        a.order = q.row_order; a.tmask = q.tile_mask; a.tbase = q.tile_base;
        total += q.work > 0 ? (double)q.work : 1.0;
    }
    std::vector<int32_t> nb((size_t)groups), map;
    int64_t used = 0;
    for (int i = 0; i < groups; ++i) {
        const double w = g[i].work > 0 ? (double)g[i].work : 1.0;
        int64_t b = (int64_t)(w / total * (cus - groups)) + 1;
        if (b > ntiles) b = ntiles;
        nb[i] = (int32_t)b;
        used += b;
    }
    map.reserve((size_t)used * 3);
    for (int32_t b = 0, more = 1; more; ++b) {
        more = 0;
        for (int i = 0; i < groups; ++i)
            if (b < nb[i]) { map.push_back(i); map.push_back(b); map.push_back(nb[i]); more = 1; }
    }
    hipStream_t st = (hipStream_t)stream;
    char *tb = (char *)table;
    const size_t map_off = ctgcn_group_table_bytes(groups) - 3 * sizeof(int32_t) * 1024;
    HIP_TRY(ctgcn_table::upload(tb, host.data(), host.size() * sizeof(GruArgs), shadow, st));
    HIP_TRY(ctgcn_table::upload(tb + map_off, map.data(), map.size() * sizeof(int32_t), shadow ? (char *)shadow + map_off : nullptr, st));
    hipLaunchKernelGGL(gru_seq_h2_group_kernel, dim3((unsigned)used), dim3(512), 0, st, (const GruArgs *)tb, (const int32_t *)(tb + map_off));
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_layer_presplit_group_f32(int32_t groups, int64_t rows, int32_t hidden, const ctgcn_gru_layer_group_t *g, void *table, size_t table_bytes,
                                       void *shadow, void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_layer_presplit_group: only d_in = hidden = %d is built (got %d)", GRU_H, hidden);
    if (groups < 1 || groups > 1024 || !g || rows < 1) return fail(CTGCN_E_INVALID, "gru_layer_presplit_group: groups=%d rows=%lld", groups, (long long)rows);
    if (!table || (reinterpret_cast<uintptr_t>(table) & 255u) || table_bytes < ctgcn_group_table_bytes(groups))
        return fail(CTGCN_E_WORKSPACE, "gru_layer_presplit_group: table must be 256-byte aligned and hold ctgcn_group_table_bytes(groups) bytes");
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    if (cus > 1024) cus = 1024;
    std::vector<LayerArgs> host((size_t)groups);
    const int64_t ntiles = (rows + 15) / 16;
    double total = 0.0;
    for (int i = 0; i < groups; ++i) {
        const ctgcn_gru_layer_group_t &q = g[i];
        if (q.steps < 1 || !q.planes || !q.w_ih || !q.w_hh || !q.out || (reinterpret_cast<uintptr_t>(q.planes) & 255u) || !aligned16(q.w_ih) || !aligned16(q.w_hh) ||
            (reinterpret_cast<uintptr_t>(q.out) & 7u) || (q.ld_out > 0 && (q.ld_out < GRU_H || (q.ld_out & 1))) ||
            (q.row_order == nullptr) != (q.tile_mask == nullptr) || (q.row_order && q.steps > 32))
            return fail(CTGCN_E_INVALID, "gru_layer_presplit_group: bad arguments of group %d", i);
        LayerArgs &a = host[i];
        std::memset((void *)&a, 0, sizeof a);
        a.rows = rows; a.steps = q.steps; a.x = nullptr; a.ldx = GRU_H; a.wih = q.w_ih; a.whh = q.w_hh; a.bias_gi = q.bias_gi; a.bhn = q.b_hn;
        a.gamma = q.ln_weight; a.beta = q.ln_bias; a.eps = q.ln_eps; a.out = q.out; a.ldo = q.ld_out > 0 ? q.ld_out : GRU_H;
        const size_t nrow = (size_t)rows * q.steps;
        a.xp1 = (const _Float16 *)q.planes; a.xp2 = a.xp1 + nrow * GRU_H; a.xps = (const float *)(a.xp2 + nrow * GRU_H);
        a.order = q.row_order; a.tmask = q.tile_mask;
        total += q.work > 0 ? (double)q.work : 1.0;
    }
    // blocks per snapshot in proportion to its work (cumulative snapshots grow with t), at least one, at most its tiles; <= one block per CU in total
    std::vector<int32_t> nb((size_t)groups), map;
    int64_t used = 0;
    if (groups > cus) return fail(CTGCN_E_UNSUPPORTED, "gru_layer_presplit_group: more snapshots (%d) than CUs (%d): split the window", groups, cus);
    for (int i = 0; i < groups; ++i) {
        const double w = g[i].work > 0 ? (double)g[i].work : 1.0;
        int64_t b = (int64_t)(w / total * (cus - groups)) + 1;
        if (b > ntiles) b = ntiles;
        nb[i] = (int32_t)b;
        used += b;
    }
    map.reserve((size_t)used * 3);
    // interleave the snapshots' blocks (block b of every snapshot before block b + 1 of any): neighbours in dispatch order start on different snapshots
    for (int32_t b = 0, more = 1; more; ++b) {
        more = 0;
        for (int i = 0; i < groups; ++i)
            if (b < nb[i]) { map.push_back(i); map.push_back(b); map.push_back(nb[i]); more = 1; }
    }
    hipStream_t st = (hipStream_t)stream;
    char *tb = (char *)table;
    const size_t map_off = ctgcn_group_table_bytes(groups) - 3 * sizeof(int32_t) * 1024;
    // the descriptors travel as kernel arguments (ctgcn_table.h: asynchronous, capturable), or not at all when `shadow` says the table is current
    HIP_TRY(ctgcn_table::upload(tb, host.data(), host.size() * sizeof(LayerArgs), shadow, st));
    HIP_TRY(ctgcn_table::upload(tb + map_off, map.data(), map.size() * sizeof(int32_t), shadow ? (char *)shadow + map_off : nullptr, st));
    hipLaunchKernelGGL(gru_layer8_h2_group_kernel, dim3((unsigned)used), dim3(512), 0, st, (const LayerArgs *)tb, (const int32_t *)(tb + map_off));
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_seq_bwd_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gates, const float *h_seq,
                          const float *dh_seq, const float *dh_sum, const float *w_hh, float *d_gi, float *d_ghn,
                          float *bias_partial, int32_t n_partial, int split_bf16, void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_seq_bwd: hidden=%d, only %d is built", hidden, GRU_H);
    if (rows < 0 || steps < 1) return fail(CTGCN_E_INVALID, "gru_seq_bwd: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!gates || !h_seq || !w_hh || !d_gi || !d_ghn || (!dh_seq && !dh_sum)) return fail(CTGCN_E_INVALID, "gru_seq_bwd: null pointer");
    GruBwdArgs a{};
    a.rows = rows; a.steps = steps; a.gates = gates; a.hseq = h_seq; a.dh_seq = dh_seq; a.dh_sum = dh_sum; a.whh = w_hh;
    a.dgi = d_gi; a.dghn = d_ghn;
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    const int64_t ntiles = (rows + GRUB_BM - 1) / GRUB_BM;       // both variants use 32-row tiles
    int64_t blocks = ntiles < cus ? ntiles : cus;
    if (bias_partial) {
        if (n_partial < 1) return fail(CTGCN_E_INVALID, "gru_seq_bwd: n_partial=%d", n_partial);
        if (blocks > n_partial) blocks = n_partial;
        a.bias_partial = bias_partial;                            // rows >= blocks of the table stay zero
        HIP_TRY(hipMemsetAsync(bias_partial, 0, (size_t)n_partial * 4 * GRU_H * sizeof(float), (hipStream_t)stream));
    }
    if (split_bf16)
        hipLaunchKernelGGL(gru_seq_bwd_x3_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_input_proj_f32(int64_t rows, int32_t d_in, int32_t hidden, const float *x, int64_t ldx,
                             const float *w_ih, const float *bias, float *gi, int split_mode, int32_t steps_blocked, void *stream)
{
    if (hidden != GRU_H || d_in != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_input_proj: only d_in = hidden = %d is built (got %d, %d)", GRU_H, d_in, hidden);
    if (rows < 0 || ldx < d_in) return fail(CTGCN_E_INVALID, "gru_input_proj: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!x || !w_ih || !gi) return fail(CTGCN_E_INVALID, "gru_input_proj: null pointer");
    if (!aligned16(x) || !aligned16(w_ih) || (ldx % 4)) return fail(CTGCN_E_INVALID, "gru_input_proj: x / w_ih must be 16-byte aligned, ldx a multiple of 4");
    ProjArgs a{};
    a.rows = rows; a.x = x; a.ldx = ldx; a.w = w_ih; a.bias = bias; a.out = gi;
    if (steps_blocked < 0 || (steps_blocked > 0 && (split_mode != CTGCN_SPLIT_F16X2 || rows % steps_blocked)))
        return fail(CTGCN_E_INVALID, "gru_input_proj: steps_blocked=%d needs CTGCN_SPLIT_F16X2 and rows a multiple of it", steps_blocked);
    a.steps_blocked = steps_blocked;

    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    const int64_t ntiles = (rows + PJ_BM - 1) / PJ_BM;
    const int64_t blocks = ntiles < cus ? ntiles : cus;
    if (split_mode == 2)
        hipLaunchKernelGGL(gru_proj_h2_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(gru_proj_x3_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_input_grad_f32(int64_t rows, int32_t d_in, int32_t hidden, const float *d_gi, const float *w_ih, float *d_x,
                             int64_t ldx, void *stream)
{
    if (hidden != GRU_H || d_in != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_input_grad: only d_in = hidden = %d is built (got %d, %d)", GRU_H, d_in, hidden);
    if (rows < 0 || ldx < d_in) return fail(CTGCN_E_INVALID, "gru_input_grad: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!d_gi || !w_ih || !d_x) return fail(CTGCN_E_INVALID, "gru_input_grad: null pointer");
    if (!aligned16(d_gi) || !aligned16(d_x) || (ldx % 4)) return fail(CTGCN_E_INVALID, "gru_input_grad: d_gi / d_x must be 16-byte aligned, ldx a multiple of 4");
    DxArgs a{};
    a.rows = rows; a.g = d_gi; a.w = w_ih; a.out = d_x; a.ldo = ldx;
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    const int64_t ntiles = (rows + GBX_BM - 1) / GBX_BM;
    const int64_t blocks = ntiles < cus ? ntiles : cus;
    hipLaunchKernelGGL(gru_dx_x3_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_weight_grad_f32(int64_t rows, int32_t steps, int32_t hidden, const float *g01, int64_t ldg01, const float *g2,
                              int64_t ldg2, const float *x, int64_t ldx, int shift_steps, float *partial, int32_t n_pairs,
                              int accumulate, void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "gru_weight_grad: hidden=%d, only %d is built", hidden, GRU_H);
    if (rows < 0 || steps < 1 || ldg01 < 2 * GRU_H || ldg2 < GRU_H || ldx < GRU_H) return fail(CTGCN_E_INVALID, "gru_weight_grad: bad sizes");
    if (n_pairs < 8 || (n_pairs % 8)) return fail(CTGCN_E_INVALID, "gru_weight_grad: n_pairs=%d must be a positive multiple of 8", n_pairs);
    if (!partial || (rows > 0 && (!g01 || !g2 || !x))) return fail(CTGCN_E_INVALID, "gru_weight_grad: null pointer");
    DwArgs a{};
    a.rows = rows; a.steps = steps; a.shift = shift_steps ? 1 : 0; a.g01 = g01; a.ldg01 = ldg01; a.g2 = g2; a.ldg2 = ldg2;
    a.x = x; a.ldx = ldx; a.partial = partial; a.pairs = n_pairs; a.accumulate = accumulate ? 1 : 0;
    hipLaunchKernelGGL(gru_dw_x3_kernel, dim3((unsigned)(2 * n_pairs)), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_lstm_seq_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gi, const float *w_hh,
                       const float *ln_weight, const float *ln_bias, float ln_eps, int reduce_sum, float *out,
                       float *gates_out, void *stream)
{
    if (gates_out && (reduce_sum || ln_weight)) return fail(CTGCN_E_INVALID, "lstm_seq: gates_out needs reduce_sum == 0 and no LayerNorm (raw h sequence)");
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "lstm_seq: hidden=%d, only %d is built", hidden, GRU_H);
    if (rows < 0 || steps < 1) return fail(CTGCN_E_INVALID, "lstm_seq: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!gi || !w_hh || !out) return fail(CTGCN_E_INVALID, "lstm_seq: null pointer");
    if (!aligned16(w_hh) || (reinterpret_cast<uintptr_t>(out) & 7u)) return fail(CTGCN_E_INVALID, "lstm_seq: w_hh must be 16-byte aligned, out 8-byte aligned");
    GruArgs a{};
    a.rows = rows; a.steps = steps; a.gi = gi; a.whh = w_hh; a.bhn = nullptr; a.gamma = ln_weight; a.beta = ln_bias;
    a.eps = ln_eps; a.reduce_sum = reduce_sum ? 1 : 0; a.out = out; a.gates = gates_out; a.ldo = GRU_H;
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    const int64_t ntiles = (rows + LSTM_BM - 1) / LSTM_BM;
    const int64_t blocks = ntiles < cus ? ntiles : cus;
    if (gates_out)
        hipLaunchKernelGGL((lstm_seq_kernel<false, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else if (a.reduce_sum)
        hipLaunchKernelGGL(lstm_seq_kernel<true>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(lstm_seq_kernel<false>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_lstm_seq_bwd_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gates, const float *dh_seq, const float *dh_sum,
                           const float *w_hh, float *d_gi, float *bias_partial, int32_t n_partial, void *stream)
{
    if (hidden != GRU_H) return fail(CTGCN_E_UNSUPPORTED, "lstm_seq_bwd: hidden=%d, only %d is built", hidden, GRU_H);
    if (rows < 0 || steps < 1) return fail(CTGCN_E_INVALID, "lstm_seq_bwd: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!gates || !w_hh || !d_gi || (!dh_seq && !dh_sum)) return fail(CTGCN_E_INVALID, "lstm_seq_bwd: null pointer");
    if (bias_partial && n_partial < 1) return fail(CTGCN_E_INVALID, "lstm_seq_bwd: n_partial must be >= 1");
    LstmBwdArgs a{};
    a.rows = rows; a.steps = steps; a.gates = gates; a.dh_seq = dh_seq; a.dh_sum = dh_sum; a.whh = w_hh; a.dgi = d_gi; a.bias_partial = nullptr;
    int dev = 0, cus = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    cus = persistent_cus(cus);
    const int64_t ntiles = (rows + LSTMB_BM - 1) / LSTMB_BM;
    int64_t blocks = ntiles < cus ? ntiles : cus;
    if (bias_partial) {
        if (blocks > n_partial) blocks = n_partial;
        a.bias_partial = bias_partial;                            // rows >= blocks of the table stay zero
        HIP_TRY(hipMemsetAsync(bias_partial, 0, (size_t)n_partial * 4 * GRU_H * sizeof(float), (hipStream_t)stream));
    }
    const size_t lds = sizeof(float) * 2 * LSTMB_BM * LSTMB_PITCH;
    (void)lds;
    hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return CTGCN_OK;
}

}  // extern "C"
