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
//   split_rows_h2_kernel   fp32 rows -> per-row scale + the two fp16 planes [rows, Kp] (Kp = K rounded up to 64, zero padded).
//                          One pass over X (read 4 B, write 4 B per element); W's planes are rebuilt per call (tiny).
//   gemm_h2_kernel         planes -> Y: 128 x 128 block tile, 4 waves (2 x 2) of 64 x 64, k steps of 32, operands staged
//                          through LDS (two stages; LDS stores of step k+1 and global loads of step k+3 interleaved
//                          with the MFMAs of step k).  Epilogue: acc·sx[m]·sw[n] + bias[n], 128-byte row segments.
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
typedef f4v f4u __attribute__((aligned(4)));            // a float4 at a 4-byte aligned address
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
                                                             _Float16 *__restrict__ p1, _Float16 *__restrict__ p2, float *__restrict__ scale,
                                                             const int32_t *__restrict__ group_map, int32_t group, float residual_scale)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *src = x + row * ldx;
    // destination row: the same, or (hub rows of the fused aggregation) group g of `group` consecutive source rows -> group group_map[g],
    // or (group < 0) group_map[row] itself, a negative entry meaning "this row is not wanted" (a repeated row under a row plan)
    const int64_t drow = !group_map ? row : (group < 0 ? (int64_t)group_map[row] : (int64_t)group_map[row / group] * group + row % group);
    if (drow < 0) return;
    float m = 0.f;
    constexpr int HOLD = 8;                               // float4 per lane kept in registers: rows up to 2048 columns are read ONCE
    f4v keep[HOLD];
    const bool held = K <= HOLD * 256;
    // VEC: 16-byte aligned rows, K % 4 == 0.  Otherwise (K = 1737: rows are only 4-byte aligned) the same 16-byte loads from 4-byte
    // aligned addresses — gfx950 serves them (unaligned access mode), the memory pipeline splits the ones that straddle a line — and
    // the last K % 4 columns of a row with scalar loads.  (The first version read and wrote element by element: 2-byte stores, 0.38 ms
    // for the 60 730 x 1737 degree features of the Facebook-like CTGCN-S window, as long as the GEMM behind it.)
    auto load4 = [&](int k) -> f4v {
        if (VEC) return *(const f4v *)(src + k);
        if (k + 4 <= K) return *(const f4u *)(src + k);
        f4v v = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = src[k + j];
        return v;
    };
    if (held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) {
            const int k = lane * 4 + i * 256;
            keep[i] = k < K ? load4(k) : f4v{0.f, 0.f, 0.f, 0.f};
            m = fmaxf(m, fmaxf(fmaxf(fabsf(keep[i][0]), fabsf(keep[i][1])), fmaxf(fabsf(keep[i][2]), fabsf(keep[i][3]))));
        }
    } else {
        for (int k = lane * 4; k < K; k += 256) {
            const f4v v = load4(k);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s, inv;
    h2_scale(m, s, inv);
    if (lane == 0) scale[drow] = s;
    _Float16 *d1 = p1 + drow * Kp, *d2 = p2 + drow * Kp;      // Kp is a multiple of 64: plane rows are 16-byte aligned whatever K is
    if (held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) {
            const int k = lane * 4 + i * 256;
            if (k < Kp) {
                h4v a, b;                                 // columns >= K were loaded as zeros
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xs = keep[i][j] * inv;
                    a[j] = (_Float16)xs;
                    b[j] = (_Float16)((xs - (float)a[j]) * residual_scale);
                }
                *(h4v *)(d1 + k) = a;
                *(h4v *)(d2 + k) = b;
            }
        }
    } else {
        for (int k = lane * 4; k < Kp; k += 256) {
            h4v a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
            if (k < K) {
                const f4v v = load4(k);                   // second read: L2 / MALL hit
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xs = v[j] * inv;
                    a[j] = (_Float16)xs;
                    b[j] = (_Float16)((xs - (float)a[j]) * residual_scale);
                }
            }
            *(h4v *)(d1 + k) = a;
            *(h4v *)(d2 + k) = b;
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
    int32_t act;                             // 0: none, 1: SELU (layers.py:103-104 F.selu after each Linear of an 'N' MLP) applied to y
    float *y;
    int64_t ldy;
    int64_t mtiles;
    int32_t ntiles;
    // chained layers (round 4, ctgcn_linear_planes_f32 with x_scale_blocks > 1 / planes out): the A operand's scales are per (row, block of
    // 128 k) — sa[m * sa_blocks + k / 128] — as a producer GEMM's 128-column tiles write them; the accumulators are rescaled (exactly: the
    // scales are powers of two) where the block changes.  o1 != null: instead of y the epilogue writes the NEXT layer's operand: planes
    // o1 / o2 [M, okp] and scales os [M, ntiles], this block's 128 columns under one scale per row.
    int32_t sa_blocks;
    _Float16 *o1, *o2;
    float *os;
    int32_t okp;
#ifdef CTGCN_GEMM_TIMELINE
    unsigned long long *timeline;   // diagnostic build (-DCTGCN_GEMM_TIMELINE): 8 words per block, see tools/gemm_timeline.py
#endif
};

// Epilogue of both GEMM kernels: y = acc·sa[m]·sb[n] + bias[n] for the wave's NI x NJ tiles of 32 x 32 (D layout: lane l holds column
// l & 31, rows 8 (v / 4) + 4 (l >> 5) + v % 4).  A tile that lies completely inside the matrix takes the straight-line path: all row
// scales requested together, then 16 NI NJ stores back to back.  With a bounds test around every store the compiler branches around
// each one and — not knowing which loads are still outstanding on which path — waits for vmcnt(0) before every store; stores
// count in vmcnt on gfx9, so each waited for the previous one to complete: 64 serialised write round trips, 25 us of a
// 46 us block life (per-block timeline with wall_clock64, 435 180 x 500 x 384).  Only edge tiles take the tested path.
// torch's SELU: scale (max(0, x) + min(0, alpha (exp(x) - 1))) with expm1 (no cancellation near 0)
__device__ __forceinline__ float gemm_act(float v, int act)
{
    if (act == 1) return v > 0.f ? 1.0507009873554804934193349852946f * v : (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f) * expm1f(v);
    return v;
}

// ------------------------------------------------------------------------------------------------
// "k3" operand form for a LIBRARY fp16 GEMM (round 4).  The three products of the fp16 x 2 split are one GEMM over concatenated planes:
//     x side  [ p1 | p2 | p1 ]   (rows x 3 kp halfs)        w side  [ w2 | w1 | w1 ]   (n_out x 3 kp halfs)
//     sum over 3 kp of x' w'^T = p1 w2 + p2 w1 + p1 w1      (fp32 accumulation inside the matrix cores, fp32 out)
// hipBLASLt's fp16 kernels reach 0.32 - 0.39 of the fp16 x 2 bound on the shapes of this path (tools/probes/lt_f16_probe.py: 800 - 980 TF/s
// executed with K' = 3 K) where gemm_h2_kernel holds 0.18 - 0.23; north_star leaves the dense Linear to the library.  The library has no
// per-row scale, so the scales travel to the CONSUMER of the raw accumulators: the next layer's split (this kernel: its prologue applies
// act(acc * row_scale * col_scale + bias)) or scale_bias_act_kernel.
// One wave per row.  fixed_max > 0: one scale for the whole tensor (weights: y's column scale is then a scalar).
template <bool VEC>
__global__ __launch_bounds__(256) void split_rows_k3_kernel(int64_t rows, int32_t K, int32_t Kp, const float *__restrict__ x, int64_t ldx,
                                                             const float *__restrict__ in_rscale, float in_cscale, const float *__restrict__ in_bias,
                                                             int32_t act, float fixed_max, int32_t w_order, _Float16 *__restrict__ planes,
                                                             float *__restrict__ scale)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *src = x + row * ldx;
    const bool xf = in_rscale != nullptr;                 // the input is a raw accumulator tile: v = act(acc rs cs + bias)
    const float rs = xf ? in_rscale[row] * in_cscale : 1.f;
    constexpr int HOLD = 8;
    f4v keep[HOLD];
    auto load4 = [&](int k) -> f4v {
        f4v v;
        if (VEC) v = *(const f4v *)(src + k);
        else if (k + 4 <= K) v = *(const f4u *)(src + k);
        else { v = f4v{0.f, 0.f, 0.f, 0.f}; for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = src[k + j]; }
        if (xf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (k + j < K) ? gemm_act(fmaf(v[j], rs, in_bias ? in_bias[k + j] : 0.f), act) : 0.f;
        }
        return v;
    };
    float m = 0.f;
    const bool held = K <= HOLD * 256;
    if (held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) {
            const int k = lane * 4 + i * 256;
            keep[i] = k < K ? load4(k) : f4v{0.f, 0.f, 0.f, 0.f};
            m = fmaxf(m, fmaxf(fmaxf(fabsf(keep[i][0]), fabsf(keep[i][1])), fmaxf(fabsf(keep[i][2]), fabsf(keep[i][3]))));
        }
    } else {
        for (int k = lane * 4; k < K; k += 256) {
            const f4v v = load4(k);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s, inv;
    h2_scale(fixed_max > 0.f ? fixed_max : m, s, inv);
    if (lane == 0 && !(fixed_max > 0.f)) scale[row] = s;
    if (fixed_max > 0.f && row == 0 && lane == 0) scale[0] = s;
    _Float16 *d = planes + row * (int64_t)(3 * Kp);
    // x side: hi | lo | hi;  w side: lo | hi | hi
    _Float16 *dh0 = d + (w_order ? Kp : 0), *dlo = d + (w_order ? 0 : Kp), *dh1 = d + 2 * Kp;
    auto put = [&](int k, const f4v v) {
        h4v a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xs = v[j] * inv;
            a[j] = (_Float16)xs;
            b[j] = (_Float16)(xs - (float)a[j]);
        }
        *(h4v *)(dh0 + k) = a; *(h4v *)(dlo + k) = b; *(h4v *)(dh1 + k) = a;
    };
    if (held) {
#pragma unroll
        for (int i = 0; i < HOLD; ++i) {
            const int k = lane * 4 + i * 256;
            if (k < Kp) put(k, keep[i]);                  // columns >= K were loaded as zeros
        }
    } else {
        for (int k = lane * 4; k < Kp; k += 256) put(k, k < K ? load4(k) : f4v{0.f, 0.f, 0.f, 0.f});
    }
}

// y = act(acc row_scale[m] col_scale + bias[n]): the last layer of a k3 chain (its consumer wants plain fp32 rows)
__global__ __launch_bounds__(256) void scale_bias_act_kernel(int64_t rows, int32_t n, const float *__restrict__ acc, int64_t lda,
                                                              const float *__restrict__ rscale, float cscale, const float *__restrict__ bias,
                                                              int32_t act, float *__restrict__ y, int64_t ldy)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = n / 4;                                  // host: n % 4 == 0, 16-byte aligned rows
    if (i >= rows * n4) return;
    const int64_t r = i / n4;
    const int c = (int)(i % n4) * 4;
    const float rs = rscale[r] * cscale;
    f4v v = *(const f4v *)(acc + r * lda + c);
    const f4v b = bias ? *(const f4v *)(bias + c) : f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = gemm_act(fmaf(v[j], rs, b[j]), act);
    *(f4v *)(y + r * ldy + c) = v;
}

template <int NI, int NJ>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, const f16v (&acc)[NI][NJ], int64_t mrow0, int ncol0, int lane, bool full)
{
    const int64_t mbase = mrow0 + 4 * (lane >> 5);
    const int nbase = ncol0 + (lane & 31);
    if (full) {
        float sc[NI][16], sb[NJ], bs[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) sc[i][v] = a.sa[(mbase + i * 32 + 8 * (v / 4) + (v % 4)) * a.sa_blocks + (a.sa_blocks - 1)];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            sb[j] = a.sb[nbase + j * 32];
            bs[j] = a.bias ? a.bias[nbase + j * 32] : 0.f;
        }
        float *yp = a.y + mbase * a.ldy + nbase;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float *row = yp + (int64_t)(i * 32 + 8 * (v / 4) + (v % 4)) * a.ldy;
#pragma unroll
                for (int j = 0; j < NJ; ++j) row[j * 32] = gemm_act(fmaf(acc[i][j][v], sc[i][v] * sb[j], bs[j]), a.act);
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nbase + j * 32;
        if (n >= a.N) continue;
        const float sb = a.sb[n], bs = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t m = mbase + i * 32 + 8 * (v / 4) + (v % 4);
                if (m < a.M) a.y[m * a.ldy + n] = gemm_act(fmaf(acc[i][j][v], a.sa[m * a.sa_blocks + (a.sa_blocks - 1)] * sb, bs), a.act);
            }
    }
}

// Epilogue of a chained layer: the block's 128 x 128 tile of act(acc sa sb + bias) leaves as the next GEMM's operand — per row ONE power-of-two
// scale for the tile's 128 columns + two fp16 planes (split_rows_h2_kernel's split under a per-(row, 128-column block) scale).  The tile goes
// through LDS in two halves of 64 rows (the k loop's stages are free by then): the waves of a half write their 64 x 64 fp32 tiles, then thread
// (row, quarter) takes 32 columns of a row — row maximum across the row's four threads, scale, split, 2 x 64 bytes out.
__device__ __forceinline__ void gemm_epilogue_planes(const GemmArgs &a, const f16v (&acc)[2][2], float *tile /* [64][132] */, int64_t m0, int n0, int nt,
                                                     int wm, int wn, int tid, int lane)
{
    constexpr int TP = 132;
    const int nb = (lane & 31);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                  // LDS free: the k loop's last reads (half 0) / the previous half's reads are done
        if (wm == half) {
            const int64_t mbase = m0 + wm * 64 + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + nb;
                const bool live = n < a.N;
                const float sb = live ? a.sb[n] : 0.f, bs = (live && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int r = i * 32 + 8 * (v / 4) + (v % 4);
                        const int64_t m = min(mbase + r, a.M - 1);
                        const float y = live ? gemm_act(fmaf(acc[i][j][v], a.sa[m * a.sa_blocks + (a.sa_blocks - 1)] * sb, bs), a.act) : 0.f;
                        tile[(r + 4 * (lane >> 5)) * TP + wn * 64 + j * 32 + nb] = y;
                    }
            }
        }
        __syncthreads();
        const int r = tid >> 2, q = tid & 3;              // 64 rows x 4 quarters of 32 columns
        const int64_t m = m0 + half * 64 + r;
        f4v v[8];
        float mx = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            v[c] = *(const f4v *)(&tile[r * TP + q * 32 + 4 * c]);
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[c][0]), fabsf(v[c][1])), fmaxf(fabsf(v[c][2]), fabsf(v[c][3]))));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1));
        mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sc, inv;
        h2_scale(mx, sc, inv);
        if (m < a.M) {
            if (q == 0) a.os[m * a.ntiles + nt] = sc;
            const int k0 = n0 + q * 32;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (k0 + 8 * c < a.okp) {                 // okp is a multiple of 64: a 8-column group is inside or outside as a whole
                    h8v p, qv;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xs = v[2 * c + e / 4][e % 4] * inv;
                        p[e] = (_Float16)xs;
                        qv[e] = (_Float16)(xs - (float)p[e]);
                    }
                    *(h8v *)(a.o1 + m * a.okp + k0 + 8 * c) = p;
                    *(h8v *)(a.o2 + m * a.okp + k0 + 8 * c) = qv;
                }
            }
        }
    }
}

// 128 x 128 block tile, 4 waves (2 x 2) of 64 x 64 (2 x 2 MFMA tiles of 32 x 32), k steps of 32 = two MFMA k slabs of 16.
// One wave is software-pipelined at half-step granularity (issue order pinned with sched_barrier; left alone the compiler
// regroups the loop by instruction kind and reads, MFMAs, stores and loads of a k step run one after the other):
//   phase A   request the fragments of slab 1 (tile kt); 12 MFMAs on slab 0, one ds_write of tile kt+1 (other LDS stage)
//             behind each of the first eight;  barrier (tile kt+1 is complete, stage kt&1 will not be read again)
//   phase B   request the fragments of slab 0 of tile kt+1 into the registers phase A just finished with; 12 MFMAs on
//             slab 1, one global load of tile kt+3 behind each of the first eight.
// One barrier per k step, two LDS stages, one set of fragment registers; global loads land 1.5 steps after their issue.
// Where the time goes (435 180 x 500 x 384; per-block timeline of the -DCTGCN_GEMM_TIMELINE build, tools/gemm_timeline.py,
// profiles/r02_gemm_timeline.txt.gz): kernel 0.77 ms, a block lives 35.5 us = prologue 5.5 (first loads, exposed) + k loop 25.0
// (16 steps; the SIMD's MFMAs of two resident blocks need 11.7 us at the 2.1 GHz the chip holds here) + epilogue 5.1; the
// first epilogue (a bounds test around every store) took 25 us of a 46 us life — see gemm_epilogue.  rocprofv3 counters of the
// kernel: matrix pipe busy 36 % of CU cycles, LDS 25 % (no bank conflicts), HBM traffic = compulsory (0.96 GB read: the three N
// tiles of a panel share its A tile in the XCD's L2, 109.8 M L2 requests, 88 % hits), average L1->L2 read latency 296 cycles.
// The matrix pipe itself sustains 1.93 PFLOP/s on v_mfma_f32_32x32x16_f16 with real operands (tools/probes/mfma_peak_probe.hip),
// not the 2.5 of the data sheet.  Measured and NOT faster: a 256 x 128 tile with 64 x 128 wave tiles at one wave per SIMD (1.09 vs
// 1.08 ms per projection, slower on the short MLP shapes), tile-contiguous global addresses (-6 %), staggered block starts (0).
// CHAIN: the layer-chain form (per-block A scales and / or planes out); the plain form keeps round 2's code and registers untouched
template <bool CHAIN, bool DMA = false>
__global__ __launch_bounds__(256, 2) void gemm_h2_kernel(const GemmArgs a)
{
    constexpr int BM = 128;
    __shared__ _Float16 As[2][2][BM][BKP];                // [stage][plane][row][k]
    __shared__ _Float16 Bs[2][2][BN][BKP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive block ids go round-robin over the 8 XCDs; the N tiles of M panel p all get p % 8
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t q = b >> 3;
    const int nt = (int)(q % a.ntiles);
    const int64_t mp = (q / a.ntiles) * 8 + xcd;
    if (mp >= a.mtiles) return;
#ifdef CTGCN_GEMM_TIMELINE
    const unsigned long long T0 = wall_clock64();
    unsigned long long T1 = 0, T2 = 0;
#endif
    const int64_t m0 = mp * BM;
    const int n0 = nt * BN;

    // staging role: a plane tile is rows x 4 segments of 16 B; thread -> rows (tid >> 2) + 64 r, segment tid & 3
    const int lr = tid >> 2, ls = (tid & 3) * 8;
    const int nk = a.Kp / BK;                             // even (Kp is a multiple of 64)
    const _Float16 *gp[8];                                // the thread's 8 sources: A plane 0/1 rows lr, lr+64; B likewise
    _Float16 *lp[8];                                      // and their LDS destinations in stage 0
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int64_t ra = min(m0 + lr + 64 * r, a.M - 1), rb = min((int64_t)n0 + lr + 64 * r, (int64_t)a.N - 1);
        const int sw = swz(lr + 64 * r, tid & 3);
        if (DMA) {
            // direct-to-LDS loads: a wave instruction fills 16 rows x 64 B = 1 KB of LDS linearly (lane l -> byte 16 l = row l / 4, stored segment
            // l % 4), so the swizzle moves to the GLOBAL side: the lane fetches the segment that belongs at its position (the XOR is its own inverse)
            gp[r] = a.a1 + ra * a.Kp + sw;      lp[r] = &As[0][0][16 * wave + 64 * r][0];
            gp[2 + r] = a.a2 + ra * a.Kp + sw;  lp[2 + r] = &As[0][1][16 * wave + 64 * r][0];
            gp[4 + r] = a.b1 + rb * a.Kp + sw;  lp[4 + r] = &Bs[0][0][16 * wave + 64 * r][0];
            gp[6 + r] = a.b2 + rb * a.Kp + sw;  lp[6 + r] = &Bs[0][1][16 * wave + 64 * r][0];
        } else {
            gp[r] = a.a1 + ra * a.Kp + ls;      lp[r] = &As[0][0][lr + 64 * r][sw];
            gp[2 + r] = a.a2 + ra * a.Kp + ls;  lp[2 + r] = &As[0][1][lr + 64 * r][sw];
            gp[4 + r] = a.b1 + rb * a.Kp + ls;  lp[4 + r] = &Bs[0][0][lr + 64 * r][sw];
            gp[6 + r] = a.b2 + rb * a.Kp + ls;  lp[6 + r] = &Bs[0][1][lr + 64 * r][sw];
        }
    }
    constexpr int A_STAGE = 2 * BM * BKP, B_STAGE = 2 * BN * BKP;          // halfs per LDS stage
    // Two register sets of 8 x 16 B: tile kt+1 waits in one while tiles kt+2 / kt+3 are in flight.  Loads are never
    // conditional: past the last k step the last tile is requested again and dropped (behind a branch the compiler cannot
    // count the outstanding loads and waits for vmcnt(0) before every LDS store: the distance silently becomes one step).
    h8v gr[2][8];
#ifndef CTGCN_GEMM_ABLATE
#define CTGCN_GEMM_ABLATE 0          // diagnostic builds (WRONG results): 1 no LDS stores in the k loop, 2 no global loads in the k loop, 3 no barrier in the k loop,
#endif                               // 4 no MFMA, 5 no fragment reads in the k loop
    auto gload1 = [&](int kt, int s, int i) { if (CTGCN_GEMM_ABLATE == 2 && kt > 2) return; gr[s][i] = *(const h8v *)(gp[i] + (int64_t)min(kt, nk - 1) * BK); };
    auto lstore1 = [&](int st, int s, int i) { if (CTGCN_GEMM_ABLATE == 1) { asm volatile("" :: "v"(gr[s][i])); return; } *(h8v *)(lp[i] + st * (i < 4 ? A_STAGE : B_STAGE)) = gr[s][i]; };

    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    // MFMA 32x32x16 operand layout: lane l holds row (l & 31), k = 8 (l >> 5) .. + 7 of a 32 x 16 slab
    const int fr = lane & 31;
    const _Float16 *fpa[2][2], *fpb[2][2];                // [slab][tile] -> plane 0 of stage 0 (plane 1: + BM*BKP, stage 1: + *_STAGE)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra = wm * 64 + t * 32 + fr, rb = wn * 64 + t * 32 + fr;
            fpa[kk][t] = &As[0][0][ra][swz(ra, kk * 2 + (lane >> 5))];
            fpb[kk][t] = &Bs[0][0][rb][swz(rb, kk * 2 + (lane >> 5))];
        }
    h8v fa[2][2][2], fb[2][2][2];                         // [slab][tile][plane]
    auto fread = [&](int st, int kk) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                fa[kk][t][p] = *(const h8v *)(fpa[kk][t] + st * A_STAGE + p * (BM * BKP));
                fb[kk][t][p] = *(const h8v *)(fpb[kk][t] + st * B_STAGE + p * (BN * BKP));
            }
    };
    // the 12 MFMAs of one slab, small terms first (x1·w2, x2·w1, then x1·w1); after MFMA number t < 8 runs side(t)
    auto slab = [&](int kk, auto side) {
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const int term = t >> 2, i = (t >> 1) & 1, j = t & 1;
            if (CTGCN_GEMM_ABLATE != 4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][i][term == 1 ? 1 : 0], fb[kk][j][term == 0 ? 1 : 0], acc[i][j], 0, 0, 0);
            else asm volatile("" :: "v"(fa[kk][i][term == 1 ? 1 : 0]), "v"(fb[kk][j][term == 0 ? 1 : 0]));
            __builtin_amdgcn_sched_barrier(0);
            if (t < 8) side(t);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // DMA: tile kt -> LDS stage kt & 1 without passing through registers (global_load_lds_dwordx4; the request counts in vmcnt)
    auto dma1 = [&](int kt, int i) {
        const int st = kt & 1;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(gp[i] + (int64_t)min(kt, nk - 1) * BK),
                                         (void __attribute__((address_space(3))) *)(lp[i] + st * (i < 4 ? A_STAGE : B_STAGE)), 16, 0, 0);
    };
    // one k step, DMA form: tile kt is complete in stage kt & 1 (slab-0 fragments in registers), tile kt+1 is landing in the other stage.
    //   phase A   fragments of slab 1; 12 MFMAs on slab 0; wait for this wave's requests of tile kt+1; barrier (tile kt+1 complete, stage kt&1 read)
    //   phase B   fragments of slab 0 of tile kt+1; 12 MFMAs on slab 1 with the eight requests of tile kt+2 (into stage kt&1) behind the first eight
    auto step_dma = [&](int kt) {
        const int st = kt & 1;
        // the fragment reads of the NEXT slab go out behind the first MFMA of this one: in front of it the compiler's s_waitcnt lgkmcnt(0)
        // for this slab's fragments (read a phase ago, long complete) would also wait for them — an LDS latency per phase
        slab(0, [&](int t) { if (t == 0) fread(st, 1); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        slab(1, [&](int t) { if (t == 0) fread(st ^ 1, 0); dma1(kt + 2, t); });
    };
    // one k step: tile kt is in LDS stage kt & 1 and its slab-0 fragments are in registers; register set s holds tile kt+1
    auto step = [&](int kt, int s) {
        const int st = kt & 1;
        if (CTGCN_GEMM_ABLATE != 5 || kt == 0) fread(st, 1);
        __builtin_amdgcn_sched_barrier(0);
        slab(0, [&](int t) { lstore1(st ^ 1, s, t); });   // stage st^1 was last read before the barrier of step kt-1
        if (CTGCN_GEMM_ABLATE != 3) __syncthreads();
        if (CTGCN_GEMM_ABLATE != 5 || kt == 0) fread(st ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        slab(1, [&](int t) { gload1(kt + 3, s, t); });
    };

    if constexpr (DMA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dma1(0, i);
#pragma unroll
        for (int i = 0; i < 8; ++i) dma1(1, i);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile 0 is in (requests retire in order)
        __syncthreads();
        fread(0, 0);
        for (int kt = 0; kt < nk; ++kt) step_dma(kt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dropped requests past the last tile must not land in the epilogue's tile
        gemm_epilogue<2, 2>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, m0 + BM <= a.M && n0 + BN <= a.N);
        return;
    }
    // prologue: tiles 0 and 1 are requested together (one exposed memory latency, not two), tile 2 as soon as set 0 is in LDS
#pragma unroll
    for (int i = 0; i < 8; ++i) gload1(0, 0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) gload1(1, 1, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) lstore1(0, 0, i);
#pragma unroll
    for (int i = 0; i < 8; ++i) gload1(2, 0, i);
    __syncthreads();
    fread(0, 0);
#ifdef CTGCN_GEMM_TIMELINE
    T1 = wall_clock64();
#endif
    if (CHAIN && a.sa_blocks > 1) {
        // A operand scaled per (row, block of 128 k = 4 k steps): entering block b the accumulators change units, acc *= s[m][b-1] / s[m][b]
        // (powers of two: exact).  D layout: lane l holds rows 8 (v / 4) + 4 (l >> 5) + v % 4 of each 32-row tile.
        const int64_t mrow = m0 + wm * 64 + 4 * (lane >> 5);
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt > 0 && (kt & 3) == 0) {
                const int b = kt >> 2;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int64_t m = min(mrow + i * 32 + 8 * (v / 4) + (v % 4), a.M - 1);
                        const float ratio = a.sa[m * a.sa_blocks + b - 1] * __builtin_amdgcn_rcpf(a.sa[m * a.sa_blocks + b]);
                        acc[i][0][v] *= ratio;
                        acc[i][1][v] *= ratio;
                        if ((v & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four rows at a time: hoisting all 64 scale loads spills the k loop
                    }
            }
            step(kt, 1);
            step(kt + 1, 0);
        }
    } else {
        for (int kt = 0; kt < nk; kt += 2) {              // unrolled by two: each register set keeps its registers
            step(kt, 1);
            step(kt + 1, 0);
        }
    }

#ifdef CTGCN_GEMM_TIMELINE
    T2 = wall_clock64();
#endif
    if (CHAIN && a.o1) gemm_epilogue_planes(a, acc, (float *)&As[0][0][0][0], m0, n0, nt, wm, wn, tid, lane);
    else gemm_epilogue<2, 2>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, m0 + BM <= a.M && n0 + BN <= a.N);
#ifdef CTGCN_GEMM_TIMELINE
    if (a.timeline && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)");           // the block's stores have left
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *d = a.timeline + (size_t)blockIdx.x * 8;
        d[0] = T0; d[1] = T1; d[2] = T2; d[3] = wall_clock64(); d[4] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

// gemm_h2_wide_kernel (round 4 experiment, CTGCN_GEMM_WIDE=1): 256 x 128 block tile, EIGHT waves (4 x 2 of 64 x 64), k steps of 32, THREE LDS
// stages filled by global_load_lds_dwordx4 — a request is in flight for two whole k steps (gemm_h2_kernel: one to one and a half), 96 KB per CU
// instead of 64.  Why: in gemm_h2_kernel the operand bytes and the MFMAs ADD (profiles/r04_gemm_ablation_dma.txt); Little's law on the A planes
// (17.6 KB/us per CU at 4.5 TB/s x ~3 us of loaded latency) asks for ~53 KB in flight per CU, which two blocks of one step each only just hold.
// Same MFMA order per accumulator as gemm_h2_kernel: bit-identical results.  One block per CU (144 KB of LDS).
__global__ __launch_bounds__(512, 2) void gemm_h2_wide_kernel(const GemmArgs a)
{
    constexpr int BM = 256, NST = 3;
    constexpr int ROWS = BM + BN;                         // plane rows of one stage: A rows 0..255, then B rows
    __shared__ _Float16 Ls[NST][2][ROWS][BKP];            // [stage][plane][row][k]: 3 x 2 x 384 x 64 B = 144 KB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const int64_t q = b >> 3;
    const int nt = (int)(q % a.ntiles);
    const int64_t mp = (q / a.ntiles) * 8 + xcd;
    if (mp >= a.mtiles) return;
    const int64_t m0 = mp * BM;
    const int n0 = nt * BN;
    const int nk = a.Kp / BK;
    // staging: a wave instruction fills 16 rows x 64 B of one plane (lane l -> row l / 4, stored segment l % 4; the swizzle sits in the global
    // address).  24 row groups x 2 planes = 48 instructions per stage, 6 per wave: groups wave, wave + 8, wave + 16 of both planes
    const _Float16 *gp[6];
    int lo[6];                                            // LDS offset (halfs) inside a stage
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int grp = wave + 8 * (i >> 1), plane = i & 1;
        const int row = grp * 16 + (lane >> 2);           // 0..383
        const int sw = swz(row, lane & 3);
        if (row < BM) gp[i] = (plane ? a.a2 : a.a1) + min(m0 + row, a.M - 1) * a.Kp + sw;
        else gp[i] = (plane ? a.b2 : a.b1) + min((int64_t)n0 + row - BM, (int64_t)a.N - 1) * a.Kp + sw;
        lo[i] = (plane * ROWS + grp * 16) * BKP;
    }
    constexpr int STAGE = 2 * ROWS * BKP;
    _Float16 *const l0 = &Ls[0][0][0][0];
    auto dma1 = [&](int kt, int i) {
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(gp[i] + (int64_t)min(kt, nk - 1) * BK),
                                         (void __attribute__((address_space(3))) *)(l0 + (kt % NST) * STAGE + lo[i]), 16, 0, 0);
    };
    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    const int fr = lane & 31;
    int fo_a[2][2], fo_b[2][2];                           // [slab][tile] -> offset (halfs) of plane 0 inside a stage
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ra = wm * 64 + t * 32 + fr, rb = BM + wn * 64 + t * 32 + fr;
            fo_a[kk][t] = ra * BKP + swz(ra, kk * 2 + (lane >> 5));
            fo_b[kk][t] = rb * BKP + swz(rb, kk * 2 + (lane >> 5));
        }
    h8v fa[2][2][2], fb[2][2][2];
    auto fread = [&](int st, int kk) {
        const _Float16 *sb = l0 + st * STAGE;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                fa[kk][t][p] = *(const h8v *)(sb + fo_a[kk][t] + p * (ROWS * BKP));
                fb[kk][t][p] = *(const h8v *)(sb + fo_b[kk][t] + p * (ROWS * BKP));
            }
    };
    auto slab = [&](int kk, auto side) {
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const int term = t >> 2, i = (t >> 1) & 1, j = t & 1;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][i][term == 1 ? 1 : 0], fb[kk][j][term == 0 ? 1 : 0], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            side(t);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll
    for (int i = 0; i < 6; ++i) dma1(0, i);
#pragma unroll
    for (int i = 0; i < 6; ++i) dma1(1, i);
#pragma unroll
    for (int i = 0; i < 6; ++i) dma1(2, i);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // tile 0 is in (requests retire in order)
    __syncthreads();
    fread(0, 0);
    int st = 0;                                           // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
        const int st1 = st == NST - 1 ? 0 : st + 1;
        // phase A: slab 0 of tile kt; the fragments of slab 1 go out behind the first MFMA; then tile kt+1 must be complete (this wave's six
        // requests of tile kt+2 may still be in flight) and everybody must have read stage st's slab 1 before it is refilled
        slab(0, [&](int t) { if (t == 0) fread(st, 1); });
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();
        // phase B: slab 1 of tile kt, fragments of slab 0 of tile kt+1, and the requests of tile kt+3 into the stage tile kt leaves
        slab(1, [&](int t) { if (t == 0) fread(st1, 0); if (t >= 1 && t < 7) dma1(kt + 3, t - 1); });
        st = st1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gemm_epilogue<2, 2>(a, acc, m0 + wm * 64, n0 + wn * 64, lane, m0 + BM <= a.M && n0 + BN <= a.N);
}

size_t align_up(size_t x, size_t al) { return (x + al - 1) / al * al; }

// CTGCN_GEMM_WIDE=1: gemm_h2_wide_kernel for the plain GEMMs (read per call)
bool gemm_wide_enabled() { const char *e = getenv("CTGCN_GEMM_WIDE"); return e && atoi(e) == 1; }
bool gemm_dma_enabled();
static void launch_gemm_plain(GemmArgs &g, hipStream_t st)
{
    if (gemm_wide_enabled()) {
        g.mtiles = (g.M + 255) / 256;
        const int64_t blocks = (g.mtiles + 7) / 8 * 8 * g.ntiles;
        hipLaunchKernelGGL(gemm_h2_wide_kernel, dim3((unsigned)blocks), dim3(512), 0, st, g);
        return;
    }
    const int64_t blocks = (g.mtiles + 7) / 8 * 8 * g.ntiles;
    if (gemm_dma_enabled()) hipLaunchKernelGGL((gemm_h2_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_h2_kernel<false, false>), dim3((unsigned)blocks), dim3(256), 0, st, g);
}

// CTGCN_GEMM_DMA=1: operand staging with global_load_lds_dwordx4 (no registers, no ds_write_b128) instead of registers + ds_write_b128.
// Measured equal (round 4, profiles/r04_gemm_ablation_dma.txt: 1.085 vs 1.096 ms on the Enron projection, windows unchanged): the k loop is not
// bound by the LDS stores' issue but by the operand bytes themselves (HBM for A, 0.35 ms of the 0.75) adding to the MFMA time instead of hiding
// under it.  Kept as a switch (read per call: tests toggle it); default off.
bool gemm_dma_enabled() { const char *e = getenv("CTGCN_GEMM_DMA"); return e && atoi(e) == 1; }

}  // namespace

extern "C" {

size_t ctgcn_linear_workspace_bytes(int64_t rows, int32_t n_out, int32_t k)
{
    if (rows < 0 || n_out < 0 || k < 0) return 0;
    const size_t kp = align_up((size_t)k, 2 * BK);        // an even number of k steps: the main loop is unrolled by two without a tail
    return align_up((size_t)rows * kp * 4 + (size_t)rows * 4, 256) + align_up((size_t)n_out * kp * 4 + (size_t)n_out * 4, 256) + 256;
}

// planes of `rows` fp32 rows: [optionally remapped] split_rows launch
static void launch_split(int64_t rows, int32_t k, int32_t kp, const float *x, int64_t ldx, _Float16 *p1, _Float16 *p2, float *scale,
                         const int32_t *group_map, int32_t group, float residual_scale, hipStream_t st)
{
    const bool vec = !(k & 3) && !(ldx & 3) && !(reinterpret_cast<uintptr_t>(x) & 15u);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (vec) hipLaunchKernelGGL(split_rows_h2_kernel<true>, grid, dim3(256), 0, st, rows, k, kp, x, ldx, p1, p2, scale, group_map, group, residual_scale);
    else hipLaunchKernelGGL(split_rows_h2_kernel<false>, grid, dim3(256), 0, st, rows, k, kp, x, ldx, p1, p2, scale, group_map, group, residual_scale);
}

// internal (ctgcn_hip.hip: hub rows of ctgcn_core_aggregate_split_f32): source row r -> plane row group_map[r / group] * group + r % group,
// or with group < 0: -> plane row group_map[r] (negative: skipped)
int ctgcn_split_rows_mapped_(int64_t rows, int32_t k, int32_t kp, const float *x, int64_t ldx, void *p1, void *p2, float *scale,
                             const int32_t *group_map, int32_t group, float residual_scale, void *stream)
{
    if (rows <= 0) return CTGCN_OK;
    launch_split(rows, k, kp, x, ldx, (_Float16 *)p1, (_Float16 *)p2, scale, group_map, group, residual_scale, (hipStream_t)stream);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

// x == nullptr: the A planes and scales are already in the workspace (ctgcn_core_aggregate_split_f32 wrote them)
static int linear_impl(int64_t rows, int32_t n_out, int32_t k, const float *x, int64_t ldx, const float *w, int64_t ldw, const float *bias, int32_t act,
                       float *y, int64_t ldy, void *workspace, size_t workspace_bytes, void *stream)
{
    if (rows < 0 || n_out < 1 || k < 1 || (x && ldx < k) || ldw < k || ldy < n_out)
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!w || !y || !workspace) return ctgcn_set_error_(CTGCN_E_INVALID, "linear: null pointer");
    if ((reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(w) & 3u) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear: x / w must be 4-byte aligned, workspace 256-byte aligned");
    if (workspace_bytes < ctgcn_linear_workspace_bytes(rows, n_out, k))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear: workspace too small (ctgcn_linear_workspace_bytes)");
    hipStream_t st = (hipStream_t)stream;
    const int32_t kp = (int32_t)align_up((size_t)k, 2 * BK);
    char *ws = (char *)workspace;
    _Float16 *a1 = (_Float16 *)ws, *a2 = a1 + (size_t)rows * kp;
    float *sa = (float *)(a2 + (size_t)rows * kp);
    char *wsb = ws + align_up((size_t)rows * kp * 4 + (size_t)rows * 4, 256);
    _Float16 *b1 = (_Float16 *)wsb, *b2 = b1 + (size_t)n_out * kp;
    float *sb = (float *)(b2 + (size_t)n_out * kp);
    if (x) launch_split(rows, k, kp, x, ldx, a1, a2, sa, nullptr, 1, 1.f, st);
    launch_split(n_out, k, kp, w, ldw, b1, b2, sb, nullptr, 1, 1.f, st);
    GemmArgs g{};
    g.M = rows; g.N = n_out; g.Kp = kp; g.a1 = a1; g.a2 = a2; g.b1 = b1; g.b2 = b2; g.sa = sa; g.sb = sb; g.bias = bias; g.act = act; g.y = y; g.ldy = ldy;
    g.sa_blocks = 1;
    g.ntiles = (n_out + BN - 1) / BN;
    g.mtiles = (rows + 127) / 128;
    const int64_t blocks = (g.mtiles + 7) / 8 * 8 * g.ntiles;
    if (blocks > 0x7fffffffLL) return ctgcn_set_error_(CTGCN_E_INVALID, "linear: too many tiles for one launch; split the rows");
#ifdef CTGCN_GEMM_TIMELINE
    static const char *tl_file = getenv("CTGCN_GEMM_TIMELINE_FILE");
    static int tl_calls = 0;
    g.timeline = nullptr;
    if (tl_file && ++tl_calls == 3) {            // the third call: clocks and caches are warm
        (void)hipMalloc(&g.timeline, (size_t)blocks * 64);
        (void)hipMemsetAsync(g.timeline, 0, (size_t)blocks * 64, st);
    }
#endif
    launch_gemm_plain(g, st);
#ifdef CTGCN_GEMM_TIMELINE
    if (g.timeline) {
        (void)hipStreamSynchronize(st);
        unsigned long long *h = (unsigned long long *)malloc((size_t)blocks * 64);
        (void)hipMemcpy(h, g.timeline, (size_t)blocks * 64, hipMemcpyDeviceToHost);
        FILE *f = fopen(tl_file, "w");
        for (int64_t i = 0; i < blocks; ++i) fprintf(f, "%lld %llu %llu %llu %llu %llu\n", (long long)i, h[i * 8], h[i * 8 + 1], h[i * 8 + 2], h[i * 8 + 3], h[i * 8 + 4]);
        fclose(f);
        free(h);
        (void)hipFree(g.timeline);
    }
#endif
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

size_t ctgcn_split_planes_bytes(int64_t rows, int32_t k)
{
    if (rows < 0 || k < 1) return 0;
    const size_t kp = align_up((size_t)k, 2 * BK);
    return align_up((size_t)rows * kp * 4 + (size_t)rows * 4, 256);
}

int ctgcn_split_rows_f32(int64_t rows, int32_t k, const float *x, int64_t ldx, void *planes, size_t planes_bytes, void *stream)
{
    if (rows < 0 || k < 1 || ldx < k) return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!x || !planes || (reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(planes) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows: x 4-byte aligned, planes 256-byte aligned");
    if (planes_bytes < ctgcn_split_planes_bytes(rows, k)) return ctgcn_set_error_(CTGCN_E_WORKSPACE, "split_rows: planes buffer too small (ctgcn_split_planes_bytes)");
    const int32_t kp = (int32_t)align_up((size_t)k, 2 * BK);
    _Float16 *p1 = (_Float16 *)planes, *p2 = p1 + (size_t)rows * kp;
    float *sc = (float *)(p2 + (size_t)rows * kp);
    launch_split(rows, k, kp, x, ldx, p1, p2, sc, nullptr, 1, 1.f, (hipStream_t)stream);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

size_t ctgcn_k3_planes_bytes(int64_t rows, int32_t k)
{
    if (rows < 0 || k < 1) return 0;
    return align_up((size_t)rows * align_up((size_t)k, 2 * BK) * 3 * sizeof(_Float16), 256);
}

int ctgcn_split_rows_k3_f32(int64_t rows, int32_t k, const float *x, int64_t ldx, const float *in_row_scale, float in_col_scale,
                            const float *in_bias, int32_t activation, float fixed_max, int32_t weight_order, void *planes, size_t planes_bytes,
                            float *scale, void *stream)
{
    if (rows < 0 || k < 1 || ldx < k) return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows_k3: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!x || !planes || !scale || (reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(planes) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows_k3: x 4-byte aligned, planes 256-byte aligned, scale required");
    if (activation != CTGCN_ACT_NONE && activation != CTGCN_ACT_SELU) return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows_k3: activation");
    if (!in_row_scale && (in_bias || activation != CTGCN_ACT_NONE)) return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows_k3: bias / activation go with in_row_scale");
    if (!(fixed_max >= 0.f)) return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows_k3: fixed_max");
    if (planes_bytes < ctgcn_k3_planes_bytes(rows, k)) return ctgcn_set_error_(CTGCN_E_WORKSPACE, "split_rows_k3: planes buffer too small (ctgcn_k3_planes_bytes)");
    const int32_t kp = (int32_t)align_up((size_t)k, 2 * BK);
    const bool vec = (ldx % 4 == 0) && (k % 4 == 0) && !(reinterpret_cast<uintptr_t>(x) & 15u);
    const dim3 grid((unsigned)((rows + 3) / 4));
    if (vec) hipLaunchKernelGGL(split_rows_k3_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, rows, k, kp, x, ldx, in_row_scale, in_col_scale, in_bias,
                                activation, fixed_max, weight_order, (_Float16 *)planes, scale);
    else hipLaunchKernelGGL(split_rows_k3_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, rows, k, kp, x, ldx, in_row_scale, in_col_scale, in_bias,
                            activation, fixed_max, weight_order, (_Float16 *)planes, scale);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_scale_bias_act_f32(int64_t rows, int32_t n, const float *acc, int64_t ld_acc, const float *row_scale, float col_scale, const float *bias,
                             int32_t activation, float *y, int64_t ldy, void *stream)
{
    if (rows < 0 || n < 4 || (n & 3) || ld_acc < n || ldy < n || (ld_acc & 3) || (ldy & 3)) return ctgcn_set_error_(CTGCN_E_INVALID, "scale_bias_act: n, ld multiples of 4");
    if (rows == 0) return CTGCN_OK;
    if (!acc || !row_scale || !y || (reinterpret_cast<uintptr_t>(acc) & 15u) || (reinterpret_cast<uintptr_t>(y) & 15u) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15u)))
        return ctgcn_set_error_(CTGCN_E_INVALID, "scale_bias_act: null or misaligned pointer");
    if (activation != CTGCN_ACT_NONE && activation != CTGCN_ACT_SELU) return ctgcn_set_error_(CTGCN_E_INVALID, "scale_bias_act: activation");
    const int64_t work = rows * (n / 4);
    hipLaunchKernelGGL(scale_bias_act_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, n, acc, ld_acc, row_scale, col_scale,
                       bias, activation, y, ldy);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

size_t ctgcn_chain_planes_bytes(int64_t rows, int32_t n_out)
{
    if (rows < 0 || n_out < 1) return 0;
    const size_t kp = align_up((size_t)n_out, 2 * BK), nblk = (size_t)(n_out + BN - 1) / BN;
    return align_up((size_t)rows * kp * 4 + (size_t)rows * nblk * 4, 256);
}

int ctgcn_linear_planes_f32(int64_t rows, int32_t n_out, int32_t k, const void *x_planes, int32_t x_scale_blocks, const void *w_planes,
                            const float *bias, int32_t activation, float *y, int64_t ldy, void *y_planes, size_t y_planes_bytes, void *stream)
{
    if (rows < 0 || n_out < 1 || k < 1 || (y && ldy < n_out) || ((y == nullptr) == (y_planes == nullptr)))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear_planes: bad sizes (exactly one of y / y_planes)");
    {
        const int32_t kp_ = (int32_t)align_up((size_t)k, 2 * BK);
        if (x_scale_blocks < 1 || (x_scale_blocks > 1 && x_scale_blocks != (kp_ + 127) / 128))
            return ctgcn_set_error_(CTGCN_E_INVALID, "linear_planes: x_scale_blocks must be 1 or ceil(kp / 128)");
        if (y_planes && ((reinterpret_cast<uintptr_t>(y_planes) & 255u) || y_planes_bytes < ctgcn_chain_planes_bytes(rows, n_out)))
            return ctgcn_set_error_(CTGCN_E_WORKSPACE, "linear_planes: y_planes must be 256-byte aligned and hold ctgcn_chain_planes_bytes(rows, n_out) bytes");
    }
    if (activation != CTGCN_ACT_NONE && activation != CTGCN_ACT_SELU) return ctgcn_set_error_(CTGCN_E_INVALID, "linear_planes: unknown activation");
    if (rows == 0) return CTGCN_OK;
    if (!x_planes || !w_planes || (reinterpret_cast<uintptr_t>(x_planes) & 255u) || (reinterpret_cast<uintptr_t>(w_planes) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear_planes: null or misaligned (256 bytes) plane buffers");
    const int32_t kp = (int32_t)align_up((size_t)k, 2 * BK);
    GemmArgs g{};
    g.M = rows; g.N = n_out; g.Kp = kp;
    g.a1 = (const _Float16 *)x_planes; g.a2 = g.a1 + (size_t)rows * kp; g.sa = (const float *)(g.a2 + (size_t)rows * kp);
    g.b1 = (const _Float16 *)w_planes; g.b2 = g.b1 + (size_t)n_out * kp; g.sb = (const float *)(g.b2 + (size_t)n_out * kp);
    g.bias = bias; g.act = activation; g.y = y; g.ldy = ldy; g.sa_blocks = x_scale_blocks;
    if (y_planes) {
        g.okp = (int32_t)align_up((size_t)n_out, 2 * BK);
        g.o1 = (_Float16 *)y_planes; g.o2 = g.o1 + (size_t)rows * g.okp; g.os = (float *)(g.o2 + (size_t)rows * g.okp);
    }
    g.ntiles = (n_out + BN - 1) / BN;
    g.mtiles = (rows + 127) / 128;
    const int64_t blocks = (g.mtiles + 7) / 8 * 8 * g.ntiles;
    if (blocks > 0x7fffffffLL) return ctgcn_set_error_(CTGCN_E_INVALID, "linear_planes: too many tiles for one launch; split the rows");
#ifdef CTGCN_GEMM_TIMELINE
    g.timeline = nullptr;
#endif
    if (x_scale_blocks > 1 || y_planes) hipLaunchKernelGGL((gemm_h2_kernel<true, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else launch_gemm_plain(g, (hipStream_t)stream);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_linear_f32(int64_t rows, int32_t n_out, int32_t k, const float *x, int64_t ldx, const float *w, int64_t ldw, const float *bias,
                     int32_t activation, float *y, int64_t ldy, void *workspace, size_t workspace_bytes, void *stream)
{
    if (rows > 0 && !x) return ctgcn_set_error_(CTGCN_E_INVALID, "linear: null pointer");
    if (activation != CTGCN_ACT_NONE && activation != CTGCN_ACT_SELU) return ctgcn_set_error_(CTGCN_E_INVALID, "linear: unknown activation");
    return linear_impl(rows, n_out, k, x, ldx, w, ldw, bias, activation, y, ldy, workspace, workspace_bytes, stream);
}

int ctgcn_linear_presplit_f32(int64_t rows, int32_t n_out, int32_t k, const float *w, int64_t ldw, const float *bias, float *y, int64_t ldy,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    return linear_impl(rows, n_out, k, nullptr, 0, w, ldw, bias, CTGCN_ACT_NONE, y, ldy, workspace, workspace_bytes, stream);
}

}  // extern "C"
