// ctgcn_gemm.hip — fp32-accurate dense  Y[M,N] = X[M,K]·W[N,K]^T + bias  on the 16-bit matrix cores of gfx950.
// Where it is used (dense steps around the CoreDiffusion aggregation whose shape the resident-weight GRU kernels do not cover):
//   * the GRU input projection of the FIRST CoreDiffusion layer of the shipped configs, d_in = hid_dim = 500 (reference
//     layers.py:59 nn.GRU(input_size=500, hidden_size=128) — its W_ih is 384 x 500),
//   * nn.Linear layers of the MLP on dense inputs (reference layers.py:95-106; CTGCN-S: 3 layers on degree features).
// Both ran as hipBLASLt fp32 GEMMs (109 TF/s, the f32-input MFMA rate class) and dominated the small-graph windows
// (Enron-like: 20 of 34 ms).  Same arithmetic as the GRU kernels' CTGCN_SPLIT_F16X2: an fp32 operand row is scaled by a
// power of two s (max|x/s| in [2^14, 2^15)) and written x/s = x1 + x2 with x1 = fp16(x/s), x2 = fp16(x/s - x1): 22
// mantissa bits; a product is x1·w1 + x1·w2 + x2·w1, three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator (small terms
// first), measured more accurate than an fp32 fma chain (tools/probes/mfma_f16x2_probe.hip).
// Two kernels:
//   split_rows_h2_kernel   fp32 rows -> per-row scale + the two fp16 planes [rows, Kp] (Kp = K rounded up to 32, zero padded).
//                          One pass over X (read 4 B, write 4 B per element); W's planes are rebuilt per call (tiny).
//   gemm_h2_kernel         planes -> Y: 128 x 128 block tile, 4 waves (2 x 2) of 64 x 64, k steps of 32, operands staged
//                          through LDS (double buffered, global loads of step k+1 in flight during the MFMAs of step k);
//                          no VALU work in the main loop.  Epilogue: acc·sx[m]·sw[n] + bias[n], 128-byte row segments.
//   Block ids are remapped so that the N tiles of one M panel run on the same XCD (its L2 then serves the panel's re-reads).
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include "../../include/ctgcn_hip.h"

extern "C" int ctgcn_set_error_(int code, const char *msg);   // defined in ctgcn_hip.hip

namespace {

#define GEMM_TRY(expr)                                                               \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            char buf[384];                                                           \
            snprintf(buf, sizeof(buf), "%s -> %s", #expr, hipGetErrorString(e_));   \
            return ctgcn_set_error_(CTGCN_E_HIP, buf);                               \
        }                                                                            \
    } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// power-of-two scale s with m/s in [2^14, 2^15) for m = max|row| (m = 0 or tiny -> a harmless huge 1/s)
__device__ __forceinline__ void h2_scale(float m, float &s, float &inv_s)
{
    int e = (int)(__float_as_uint(m) >> 23);
    e = e < 15 ? 15 : (e > 253 ? 253 : e);
    s = __uint_as_float((uint32_t)(e - 14) << 23);
    inv_s = __uint_as_float((uint32_t)(268 - e) << 23);
}

// one wave per row: max |x| -> scale; planes p1 = fp16(x/s), p2 = fp16(x/s - p1); columns [K, Kp) are zero.
// VEC: rows are 16-byte aligned and K % 4 == 0 (float4 loads); otherwise scalar loads (coalesced 4-byte, e.g. K = 1737).
template <bool VEC>
__global__ __launch_bounds__(256) void split_rows_h2_kernel(int64_t rows, int32_t K, int32_t Kp, const float *__restrict__ x, int64_t ldx,
                                                             _Float16 *__restrict__ p1, _Float16 *__restrict__ p2, float *__restrict__ scale)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *src = x + row * ldx;
    float m = 0.f;
    constexpr int HOLD = 8;                               // float4 per lane kept in registers: rows up to 2048 columns are read ONCE
    f4v keep[HOLD];
    const bool held = VEC && K <= HOLD * 256;
    if (VEC) {
        if (held) {
#pragma unroll
            for (int i = 0; i < HOLD; ++i) {
                const int k = lane * 4 + i * 256;
                keep[i] = k < K ? *(const f4v *)(src + k) : f4v{0.f, 0.f, 0.f, 0.f};
                m = fmaxf(m, fmaxf(fmaxf(fabsf(keep[i][0]), fabsf(keep[i][1])), fmaxf(fabsf(keep[i][2]), fabsf(keep[i][3]))));
            }
        } else {
            for (int k = lane * 4; k < K; k += 256) {
                const f4v v = *(const f4v *)(src + k);
                m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
        }
    } else {
        for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(src[k]));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s, inv;
    h2_scale(m, s, inv);
    if (lane == 0) scale[row] = s;
    _Float16 *d1 = p1 + row * Kp, *d2 = p2 + row * Kp;
    if (VEC && held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) {
            const int k = lane * 4 + i * 256;
            if (k < Kp) {
                h4v a, b;                                 // columns >= K were loaded as zeros
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xs = keep[i][j] * inv;
                    a[j] = (_Float16)xs;
                    b[j] = (_Float16)(xs - (float)a[j]);
                }
                *(h4v *)(d1 + k) = a;
                *(h4v *)(d2 + k) = b;
            }
        }
    } else if (VEC) {
        for (int k = lane * 4; k < Kp; k += 256) {
            h4v a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
            if (k < K) {                                  // K % 4 == 0: a float4 is entirely inside or outside
                const f4v v = *(const f4v *)(src + k);    // second read: L2 / MALL hit
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xs = v[j] * inv;
                    a[j] = (_Float16)xs;
                    b[j] = (_Float16)(xs - (float)a[j]);
                }
            }
            *(h4v *)(d1 + k) = a;
            *(h4v *)(d2 + k) = b;
        }
    } else {
        for (int k = lane; k < Kp; k += 64) {
            _Float16 a = 0, b = 0;
            if (k < K) {
                const float xs = src[k] * inv;
                a = (_Float16)xs;
                b = (_Float16)(xs - (float)a);
            }
            d1[k] = a;
            d2[k] = b;
        }
    }
}

constexpr int BN = 128, BK = 32, BKP = BK;          // LDS rows are 64 B, unpadded: 16-byte segment s of row r is stored at s ^ ((r >> 2) & 3)
// (8 consecutive lanes of a ds_read_b128 / ds_write_b128 then hit 8 distinct bank groups).  With 8 halfs of padding per row the
// double-buffered 128 x 128 tile took exactly half of the CU's 160 KB and only ONE block was resident (measured 2 waves per CU).
__device__ __forceinline__ int swz(int row, int seg) { return ((seg ^ ((row >> 2) & 3)) << 3); }

struct GemmArgs {
    int64_t M;
    int32_t N, Kp;
    const _Float16 *a1, *a2, *b1, *b2;       // planes [M, Kp] / [N, Kp]
    const float *sa, *sb, *bias;             // row scales, bias[N] or null
    float *y;
    int64_t ldy;
    int64_t mtiles;
    int32_t ntiles;
};

// WM = 32-row MFMA tiles per wave along M: 2 -> 128 x 128 block tile, LDS double buffered (one barrier per k step); the
// template also builds 4 -> 256 x 128 block tile, LDS single buffered (128 x 64 wave tiles read 12 KB of operands per 24
// MFMAs instead of 8 KB per 12), not instantiated: it needs the registers of the second load set.
// Measured (rocprofv3 SQ counters, 435 180 x 500 x 384): 0.99 ms, matrix pipe busy 31 % of CU-busy cycles, waves waiting 47 % —
// neither two blocks per CU (swizzled LDS), nor loads two k steps ahead, nor the taller wave tile moved it by more than 7 %.
template <int WM>
__global__ __launch_bounds__(256, 2) void gemm_h2_kernel(const GemmArgs a)
{
    constexpr int BM = 64 * WM;
    constexpr int ST = WM == 2 ? 2 : 1;                  // LDS stages
    __shared__ _Float16 As[ST][2][BM][BKP];
    __shared__ _Float16 Bs[ST][2][BN][BKP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive block ids go round-robin over the 8 XCDs; the N tiles of M panel p all get p % 8
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t q = b >> 3;
    const int nt = (int)(q % a.ntiles);
    const int64_t mp = (q / a.ntiles) * 8 + xcd;
    if (mp >= a.mtiles) return;
    const int64_t m0 = mp * BM;
    const int n0 = nt * BN;

    // staging role: a plane tile is rows x 4 segments of 16 B; thread -> rows (tid >> 2) + 64 r, segment tid & 3
    const int lr = tid >> 2, ls = (tid & 3) * 8;
    const int nk_ = a.Kp / BK;
    // Two register sets: the tile of k step kt+1 waits in one while the loads of kt+2 / kt+3 are in flight in both — a global
    // load issued only one k step (0.35 us of MFMAs) before its ds_write is still on its way (~1.5 us loaded latency).
    h8v ga[2][2][WM], gb[2][2][2];           // [set][plane][row group]
    auto gload = [&](int kt, int s) {
        if (kt >= nk_) return;
        const int64_t k = (int64_t)kt * BK + ls;
#pragma unroll
        for (int r = 0; r < WM; ++r) {
            const int64_t row = min(m0 + lr + 64 * r, a.M - 1);
            ga[s][0][r] = *(const h8v *)(a.a1 + row * a.Kp + k);
            ga[s][1][r] = *(const h8v *)(a.a2 + row * a.Kp + k);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t row = min((int64_t)n0 + lr + 64 * r, (int64_t)a.N - 1);
            gb[s][0][r] = *(const h8v *)(a.b1 + row * a.Kp + k);
            gb[s][1][r] = *(const h8v *)(a.b2 + row * a.Kp + k);
        }
    };
    auto lstore = [&](int st, int s) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int r = 0; r < WM; ++r) *(h8v *)(&As[st][p][lr + 64 * r][swz(lr + 64 * r, tid & 3)]) = ga[s][p][r];
#pragma unroll
            for (int r = 0; r < 2; ++r) *(h8v *)(&Bs[st][p][lr + 64 * r][swz(lr + 64 * r, tid & 3)]) = gb[s][p][r];
        }
    };

    f16v acc[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    const int nk = nk_;
    gload(0, 0);
    lstore(0, 0);
    gload(1, 1);
    gload(2, 0);
    __syncthreads();
    // MFMA 32x32x16 operand layout: lane l holds row (l & 31), k = 8 (l >> 5) .. + 7 of a 32 x 16 slab
    const int fr = lane & 31;
    auto compute = [&](int st) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8v fa[WM][2], fb[2][2];                                      // [tile][plane]
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row = wm * (32 * WM) + i * 32 + fr, off = swz(row, kk * 2 + (lane >> 5));
                fa[i][0] = *(const h8v *)(&As[st][0][row][off]);
                fa[i][1] = *(const h8v *)(&As[st][1][row][off]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + fr, off = swz(row, kk * 2 + (lane >> 5));
                fb[j][0] = *(const h8v *)(&Bs[st][0][row][off]);
                fb[j][1] = *(const h8v *)(&Bs[st][1][row][off]);
            }
            // small terms first: x1·w2, x2·w1, then x1·w1
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
        }
    };
    // one k step: compute from LDS stage st, then move the tile of step kt+1 (register set s) to LDS and request step kt+3 into s
    auto step = [&](int kt, int s) {
        const int st = ST == 2 ? (kt & 1) : 0;
        compute(st);
        if (ST == 2) {
            if (kt + 1 < nk) lstore(st ^ 1, s);                           // stage st^1 was last read in iteration kt-1 (barrier since)
            gload(kt + 3, s);
            __syncthreads();
        } else {
            if (kt + 1 < nk) {
                __syncthreads();                                          // everyone has read the stage
                lstore(0, s);
            }
            gload(kt + 3, s);
            __syncthreads();
        }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, 1);                                                      // the tile of step kt+1 (odd) sits in set 1
        if (kt + 1 < nk) step(kt + 1, 0);
    }

    // epilogue: D[i][j] of a 32 x 32 tile: lane l holds column j = l & 31, rows i = 8 (v / 4) + 4 (l >> 5) + v % 4
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= a.N) continue;
        const float sb = a.sb[n], bs = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t m = m0 + wm * (32 * WM) + i * 32 + 8 * (v / 4) + 4 * (lane >> 5) + (v % 4);
                if (m < a.M) a.y[m * a.ldy + n] = fmaf(acc[i][j][v], a.sa[m] * sb, bs);
            }
        }
    }
}

size_t align_up(size_t x, size_t al) { return (x + al - 1) / al * al; }

}  // namespace

extern "C" {

size_t ctgcn_linear_workspace_bytes(int64_t rows, int32_t n_out, int32_t k)
{
    if (rows < 0 || n_out < 0 || k < 0) return 0;
    const size_t kp = align_up((size_t)k, BK);
    return align_up((size_t)rows * kp * 4 + (size_t)rows * 4, 256) + align_up((size_t)n_out * kp * 4 + (size_t)n_out * 4, 256) + 256;
}

int ctgcn_linear_f32(int64_t rows, int32_t n_out, int32_t k, const float *x, int64_t ldx, const float *w, int64_t ldw, const float *bias,
                     float *y, int64_t ldy, void *workspace, size_t workspace_bytes, void *stream)
{
    if (rows < 0 || n_out < 1 || k < 1 || ldx < k || ldw < k || ldy < n_out)
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!x || !w || !y || !workspace) return ctgcn_set_error_(CTGCN_E_INVALID, "linear: null pointer");
    if ((reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(w) & 3u) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear: x / w must be 4-byte aligned, workspace 256-byte aligned");
    const bool vx = !(k & 3) && !(ldx & 3) && !(reinterpret_cast<uintptr_t>(x) & 15u);
    const bool vw = !(k & 3) && !(ldw & 3) && !(reinterpret_cast<uintptr_t>(w) & 15u);
    if (workspace_bytes < ctgcn_linear_workspace_bytes(rows, n_out, k))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear: workspace too small (ctgcn_linear_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    const int32_t kp = (int32_t)align_up((size_t)k, BK);
    char *ws = (char *)workspace;
    _Float16 *a1 = (_Float16 *)ws, *a2 = a1 + (size_t)rows * kp;
    float *sa = (float *)(a2 + (size_t)rows * kp);
    char *wsb = ws + align_up((size_t)rows * kp * 4 + (size_t)rows * 4, 256);
    _Float16 *b1 = (_Float16 *)wsb, *b2 = b1 + (size_t)n_out * kp;
    float *sb = (float *)(b2 + (size_t)n_out * kp);
    if (vx) hipLaunchKernelGGL(split_rows_h2_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, rows, k, kp, x, ldx, a1, a2, sa);
    else hipLaunchKernelGGL(split_rows_h2_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, rows, k, kp, x, ldx, a1, a2, sa);
    if (vw) hipLaunchKernelGGL(split_rows_h2_kernel<true>, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, st, (int64_t)n_out, k, kp, w, ldw, b1, b2, sb);
    else hipLaunchKernelGGL(split_rows_h2_kernel<false>, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, st, (int64_t)n_out, k, kp, w, ldw, b1, b2, sb);
    GemmArgs g{};
    g.M = rows; g.N = n_out; g.Kp = kp; g.a1 = a1; g.a2 = a2; g.b1 = b1; g.b2 = b2; g.sa = sa; g.sb = sb; g.bias = bias; g.y = y; g.ldy = ldy;
    g.ntiles = (n_out + BN - 1) / BN;
    constexpr int wm = 2;          // 128-row panels (a 256-row panel / 128 x 64 wave tile was measured: 7 % faster single-set, spills with two sets)
    const int bm = 64 * wm;
    g.mtiles = (rows + bm - 1) / bm;
    const int64_t blocks = (g.mtiles + 7) / 8 * 8 * g.ntiles;
    if (blocks > 0x7fffffffLL) return ctgcn_set_error_(CTGCN_E_INVALID, "linear: too many tiles for one launch; split the rows");
    hipLaunchKernelGGL(gemm_h2_kernel<wm>, dim3((unsigned)blocks), dim3(256), 0, st, g);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

}  // extern "C"
