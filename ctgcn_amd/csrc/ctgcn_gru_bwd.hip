// ctgcn_gru_bwd.hip — fused backward of the GRU with d_in = hidden = 128 (the core-axis GRU of CoreDiffusion, reference layers.py:59-62,
// and the temporal GRU of CTGCN, models.py:249-250): what autograd derives for nn.GRU + sum + the aggregation's ReLU, in TWO kernels that
// keep every product's operands on the CU.  gfx950 only.
//
// Round-3 backward per row-step: recompute (gates 2 KB out) -> gru_seq_bwd_x3 (gates in, dGI 1.5 KB + dGHn 0.5 KB out) -> gru_dx_x3 (dGI in)
// -> gru_dw_x3 twice (dGI, dGHn, x, h in) -> agg_bwd_prep (dH, H in, Z out): 14 KB of HBM traffic per row-step, every kernel at its HBM time.
// Here:
//   gru_bwd_rec_kernel  ("recurrence side")  gates + h in; backward recurrence with W_hh^T resident; dW_hh accumulated in registers over all
//                       the block's tiles (per-block partials, one deterministic reduction by the caller); dGH never leaves the CU; writes
//                       dGI summed over the steps that repeat one x (row plan) — the only intermediate: 1.5 KB per FRESH row-step.
//   gru_bwd_in_kernel   ("input side")       dGI sums + x in; dx = dGI·W_ih with W_ih^T resident, dW_ih accumulated in registers; the epilogue
//                       applies the aggregation's ReLU mask and its (double) suffix sum over the core axis and writes Z / S0 — the operands of
//                       agg_bwd_kernel — directly: the fp32 dH [N, K, d] and agg_bwd_prep_kernel's pass over it do not exist any more.
// Arithmetic: bf16 x 2 split (x = hi + lo, 16 mantissa bits, no scales: gradients have no row structure worth a scale and the weight-gradient
// products contract over ROWS, where a per-row scale cannot be factored out), three v_mfma_f32_16x16x32_bf16 per product (lo·hi, hi·lo, hi·hi),
// fp32 accumulation: relative error 2^-17 per operand against 2^-25 of round 3's bf16 x 3 with six products — half the matrix work, and the
// gradient tolerances (1e-4 of a tensor's largest entry, tests/test_gpu_models.py) are two orders above it.
// Weight-gradient products contract over the 16 rows of a tile: K = 32 of the MFMA is filled with [G_hi | G_hi] x [X_hi ; X_lo] and
// [G_lo | G_lo] x [X_hi ; 0]; the row-major planes in LDS are read TRANSPOSED with ds_read_b64_tr_b16 (no VALU transposes, no second layout).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include <hip/hip_runtime.h>

#include "../../include/ctgcn_hip.h"
#include "ctgcn_jitter.h"          // diagnostic builds (-DCTGCN_JITTER): delays around every barrier; nothing in the product

extern "C" int ctgcn_set_error_(int code, const char *msg);   // defined in ctgcn_hip.hip
extern "C" int ctgcn_persistent_cus_(int device_cus);         // "

namespace {

#define BWD_TRY(expr)                                                                \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) {                                                      \
            char buf[384];                                                           \
            snprintf(buf, sizeof(buf), "%s -> %s", #expr, hipGetErrorString(e_));   \
            return ctgcn_set_error_(CTGCN_E_HIP, buf);                               \
        }                                                                            \
    } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf4v __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

constexpr int GH = 128;
constexpr int G3 = 3 * GH;
// plane row pitches in bf16 elements: 200 / 72 dwords = 8 (mod 64).  With lane = (row, 16-byte k group) the ds_read_b128 operand reads are
// conflict-free under the hardware's 16-lane service groups (rows {0-3, 12-15} of k group g with rows {4-11} of k group g + 1), and a
// ds_read_b64_tr_b16 whose two 16-lane groups of a half-wave take row quads {4a .. 4a+3} and {4a+4 .. 4a+7} covers all 64 banks once.
constexpr int GP = G3 + 16;
constexpr int HP = GH + 16;

__device__ __forceinline__ void bf16_split2(float x, __bf16 &hi, __bf16 &lo)
{
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}
__device__ __forceinline__ void split4(const f4v v, bf4v &hi, bf4v &lo)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __bf16 a, b;
        bf16_split2(v[j], a, b);
        hi[j] = a; lo[j] = b;
    }
}
// W^T fragments of one 16-column slice: A[m = 16 tile + col][k = 32 c + 8 grp + jj] = w[k][16 tile + col]  (w is [384, 128] row-major)
// lo fragments 0 .. NL-1 go to LDS (lo_dst, 64 fragments apart), the last 12 - NL stay in registers (lo_reg)
template <int NL>
__device__ __forceinline__ void load_wt_fragments(const float *w, int tile, int col, int grp, bf8v (&hi)[12], bf8v *lo_dst, bf8v (&lo_reg)[12 - NL + 1])
{
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        bf8v lo;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            __bf16 a, b;
            bf16_split2(w[(int64_t)(c * 32 + 8 * grp + jj) * GH + tile * 16 + col], a, b);
            hi[c][jj] = a; lo[jj] = b;
        }
        if (c < NL) lo_dst[c * 64] = lo;
        else lo_reg[c - NL] = lo;
    }
}
__device__ __forceinline__ bf8v tr_pair(const __bf16 *p0, const __bf16 *p1)
{
    typedef __attribute__((address_space(3))) s4v lds_s4v;
    const s4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v *)p0);
    const s4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v *)p1);
    typedef short s8v __attribute__((ext_vector_type(8)));
    const s8v v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf8v, v);
}

// ------------------------------------------------------------------------------------------------
// The weight-gradient block shared by both kernels:  acc[ut][jt] += sum over the tile's 16 rows of G[row][16 (3 wave + ut) + m] · X[row][16 jt + n]
// G planes [2][16][GP] and X planes [2][16][HP] in LDS (hi, lo).  k slot 8 g + i of the MFMA holds row 4 (g & 1) + (i & 3) + 8 (i >> 2) of plane
// copy g >> 1: k 0..15 = the 16 rows, k 16..31 = the 16 rows again — A = [G_p | G_p], B1 = [X_hi ; X_lo], B2 = [X_hi ; 0].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void weight_grad_block(const __bf16 *Gs, const __bf16 *Xs, int wave, int lane, f4v (&acc)[3][8])
{
    const int grp = lane >> 4;
    const int r0 = 4 * (grp & 1) + ((lane & 15) >> 2), c4 = 4 * (lane & 3);
    bf8v A1[3], A2[3];
#pragma unroll
    for (int ut = 0; ut < 3; ++ut) {
        const int cu = 16 * (3 * wave + ut) + c4;
        A1[ut] = tr_pair(Gs + r0 * GP + cu, Gs + (r0 + 8) * GP + cu);
        A2[ut] = tr_pair(Gs + (16 + r0) * GP + cu, Gs + (16 + r0 + 8) * GP + cu);
    }
    const __bf16 *xb = Xs + ((grp >> 1) * 16 + r0) * HP + c4;
    const bf8v zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
        const bf8v B1 = tr_pair(xb + 16 * jt, xb + 8 * HP + 16 * jt);
        const bf8v B2 = grp < 2 ? B1 : zero8;
#pragma unroll
        for (int ut = 0; ut < 3; ++ut) {
            acc[ut][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[ut], B2, acc[ut][jt], 0, 0, 0);
            acc[ut][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[ut], B1, acc[ut][jt], 0, 0, 0);
        }
    }
}
// acc -> partial[block][384][128]: lane (col, grp) holds D[m = 4 grp + i][n = col] of tile (ut, jt).  With `accumulate` all 96 reads of the
// running sums are issued before the first add (one guarded read-modify-write per element serialises 96 memory latencies).
__device__ __forceinline__ void store_weight_grad(float *part, int wave, int lane, int accumulate, f4v (&acc)[3][8])
{
    const int col = lane & 15, grp = lane >> 4;
    float *out = part + (int64_t)blockIdx.x * G3 * GH + (16 * 3 * wave + 4 * grp) * GH + col;
    if (accumulate) {
        f4v prev[3][8];
#pragma unroll
        for (int ut = 0; ut < 3; ++ut)
#pragma unroll
            for (int jt = 0; jt < 8; ++jt)
#pragma unroll
                for (int i = 0; i < 4; ++i) prev[ut][jt][i] = out[(16 * ut + i) * GH + 16 * jt];
#pragma unroll
        for (int ut = 0; ut < 3; ++ut)
#pragma unroll
            for (int jt = 0; jt < 8; ++jt) acc[ut][jt] += prev[ut][jt];
    }
#pragma unroll
    for (int ut = 0; ut < 3; ++ut)
#pragma unroll
        for (int jt = 0; jt < 8; ++jt)
#pragma unroll
            for (int i = 0; i < 4; ++i) out[(16 * ut + i) * GH + 16 * jt] = acc[ut][jt][i];
}
// the product with the resident W^T slice: D[m = out unit 16 wave + 4 grp + i][n = row col] = sum_k W^T[m][k] G[row][k], k over the 384 gate columns
template <int NL>
__device__ __forceinline__ f4v gate_product(const __bf16 *Gs, const bf8v (&Wh)[12], const bf8v *wl, const bf8v (&lo_reg)[12 - NL + 1], int col, int grp)
{
    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    f4v a0 = zero4, a1 = zero4, a2 = zero4;
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        const bf8v gh = *(const bf8v *)(Gs + col * GP + 32 * c + 8 * grp);
        const bf8v gl = *(const bf8v *)(Gs + (16 + col) * GP + 32 * c + 8 * grp);
        const bf8v wlo = c < NL ? wl[(c < NL ? c : 0) * 64] : lo_reg[c < NL ? 0 : c - NL];
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wh[c], gl, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, gh, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wh[c], gh, a2, 0, 0, 0);
    }
    return (a0 + a1) + a2;
}

// ------------------------------------------------------------------------------------------------
// gru_bwd_rec_kernel: walks t = steps-1 .. 0 with dh = (SUM ? dh[row] : dh[row, t]) + the recurrent term carried in registers:
//   dn = dh(1-z); dz = dh(h_{t-1}-n); da_n = dn(1-n^2); da_z = dz z(1-z); da_r = da_n q r(1-r); dgh_n = da_n r
//   dh_{t-1} = dh z + (da_r, da_z, dgh_n)·W_hh            dW_hh += (da_r, da_z, dgh_n)^T h_{t-1}
//   dgi[row, t] = sum over t and the steps after it that repeat x_t (tmask bit clear) of (da_r, da_z, da_n)      (written at fresh steps only)
// 16-row tiles, eight waves, wave w owns hidden units [16w, 16w+16) of dh and rows [48w, 48w+48) of dW_hh.  A lane owns (row col, 4 units)
// — the MFMA D layout — for the whole tile.  Gates and h_{t-1} of step t-1 are requested while step t multiplies.
// LDS: 11 of the 12 lo fragments of W_hh^T (88 KB) + the planes of the step's (da_r | da_z | dgh_n) (25 KB) and h_{t-1} (9 KB), double
// buffered by step: ONE barrier per step, and the weight-gradient products of step t + 1 (off the recurrence's path) run in front of step
// t's barrier, where early waves used to wait.
// ------------------------------------------------------------------------------------------------
// step masks of the row plan: one 32-bit word per 16-row tile up to 32 steps, two (steps 0-31, 32-63) beyond (ctgcn_amd.core_adj.CoreAdj.row_plan;
// America-Air / Europe-Air depth, reference README.md:175-176).  Always carried as 64 bits here (uniform: scalar registers); no plan = all ones,
// which is also what keeps `mask >> t` defined for the plan-less 33-64 step calls (round 5 found that shift wrong in the forward kernel).
__device__ __forceinline__ uint64_t bwd_step_mask(const uint32_t *tm, int64_t tile, int S)
{
    if (!tm) return ~0ull;
    uint32_t lo, hi = 0;
    if (S > 32) { lo = tm[2 * tile]; hi = tm[2 * tile + 1]; } else lo = tm[tile];
    // the tile is uniform over the block: both halves live in scalar registers (left to the compiler one of them took a vector register
    // of gru_bwd_rec_kernel, which has none to spare: 14 spilled registers instead of 13)
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)lo) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32);
}
__device__ __forceinline__ int bwd_top_step(uint64_t mask, int S) { return 63 - __builtin_clzll(mask & (S >= 64 ? ~0ull : ((1ull << S) - 1ull))); }

struct BwdRecArgs {
    int64_t rows;
    int32_t steps;
    const float *gates;      // [rows, steps, 4, 128] r, z, n, q — or (gates3) [rows, steps, 3, 128] r, z, q with n rebuilt from h_t, h_{t-1}, z
    int32_t gates3;
    const float *hseq;       // [rows, steps, 128]
    const float *dh;         // SUM: [rows, 128]; else [rows, steps, 128]
    const float *whh;        // [384, 128]
    const uint32_t *tmask;   // per 16-row tile: bit t set = x_t is new for the tile; null = every step
    float *dgi;              // [rows, steps, 384]
    float *dw_part;          // [gridDim.x, 384, 128]
    float *dbn_part;         // [gridDim.x, 128]: column sums of dgh_n
    int32_t accumulate;
    int32_t ablate;          // diagnostic (CTGCN_BWD_ABLATE): 1 no stores, 2 no weight-gradient products, 4 no gate product, 8 loads of the first step only
};

// WIDE (round 6): 33-64 steps, two mask words per tile.  A separate instantiation: the 32-step kernels keep their 32-bit mask and their register
// allocation (the unified 64-bit form cost gru_bwd_rec_kernel a fourteenth spilled register).
template <bool SUM, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void gru_bwd_rec_kernel(const BwdRecArgs a)
{
    constexpr int NL = 11;                                 // lo fragments of W_hh^T in LDS (88 KB); the twelfth stays in registers: the planes are double buffered
    constexpr int GBUF = 2 * 16 * GP, HBUF = 2 * 16 * HP;
    __shared__ __bf16 Gs[2 * GBUF];                        // [step parity][plane][row][GP]
    __shared__ __bf16 Hs[2 * HBUF];
    __shared__ bf8v Wl[8 * NL * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int oc = wave * 16 + 4 * grp;
    const int S = a.steps;

    bf8v Wh[12], wreg[12 - NL + 1];
    load_wt_fragments<NL>(a.whh, wave, col, grp, Wh, &Wl[(wave * NL) * 64 + lane], wreg);
    const bf8v *wl = &Wl[(wave * NL) * 64 + lane];
    __syncthreads();

    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    f4v acc[3][8];
#pragma unroll
    for (int ut = 0; ut < 3; ++ut)
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) acc[ut][jt] = zero4;
    f4v bsn = zero4;

    const int64_t ntiles = (a.rows + 15) / 16;
    int pb = 0;                                            // plane buffer the next publish goes to; flips with every published step, across tiles too
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * 16;
        const int last = (int)min((int64_t)16, a.rows - row0) - 1;
        const bool valid = col <= last;
        const int64_t row = row0 + min(col, last);
        typename std::conditional<WIDE, uint64_t, uint32_t>::type tmask;
        if constexpr (WIDE) tmask = bwd_step_mask(a.tmask, tile, S);
        else tmask = a.tmask ? a.tmask[tile] : 0xffffffffu;
        f4v drec = zero4, gs0 = zero4, gs1 = zero4, gs2 = zero4;
        f4v dhs = zero4;
        if (SUM) dhs = *(const f4v *)(a.dh + row * GH + oc);
        f4v gr, gz, gn = zero4, gq, hp, dht = zero4;
        auto load_step = [&](int t) {
            const int64_t e = row * S + t;
            if (a.gates3) {
                const float *gp = a.gates + e * (3 * GH) + oc;
                gr = *(const f4v *)gp; gz = *(const f4v *)(gp + GH); gq = *(const f4v *)(gp + 2 * GH);
            } else {
                const float *gp = a.gates + e * (4 * GH) + oc;
                gr = *(const f4v *)gp; gz = *(const f4v *)(gp + GH); gn = *(const f4v *)(gp + 2 * GH); gq = *(const f4v *)(gp + 3 * GH);
            }
            hp = t > 0 ? *(const f4v *)(a.hseq + (e - 1) * GH + oc) : zero4;
            if (!SUM) dht = *(const f4v *)(a.dh + e * GH + oc);
        };
        f4v hcur = zero4;                                  // gates3: h_t of the step in hand (= the h_{t-1} the step after it loaded)
        if (a.gates3) hcur = *(const f4v *)(a.hseq + (row * S + S - 1) * GH + oc);
        load_step(S - 1);
        bool pending = false;                              // the weight-gradient products of the step before (t + 1) are still to do
        for (int t = S - 1; t >= 0; --t) {
            __bf16 *Gc = Gs + pb * GBUF, *Hc = Hs + pb * HBUF;
            f4v dh = drec + (SUM ? dhs : dht);
            if (!valid) dh = zero4;                       // rows past the end read the last row's data: they must not reach dW / db
            if (a.gates3) {
                // h_t = n + z (h_{t-1} - n)  =>  n = h_{t-1} + (h_t - h_{t-1}) / (1 - z).  Every use of n below carries a factor (1 - z)
                // (da_n, and through it da_r, dgh_n) or is the difference h_{t-1} - n = -(h_t - h_{t-1}) / (1 - z) times (1 - z) (da_z): the
                // division's error is multiplied back by what it was divided by.  1 - z below 1e-6: n := h_{t-1} (da_n is then < 1e-6 dh).
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float omz = 1.f - gz[j];
                    gn[j] = omz > 1e-6f ? hp[j] + (hcur[j] - hp[j]) * __builtin_amdgcn_rcpf(omz) : hp[j];
                }
            }
            const f4v dan = dh * (1.f - gz) * (1.f - gn * gn);
            // (h_{t-1} - n)(1 - z) = h_{t-1} - h_t exactly: with three gates da_z needs no n at all
            const f4v daz = a.gates3 ? dh * (hp - hcur) * gz : dh * (hp - gn) * gz * (1.f - gz);
            if (a.gates3) hcur = hp;
            const f4v dar = dan * gq * gr * (1.f - gr);
            const f4v dgn = dan * gr;
            drec = dh * gz;                                // direct path; the W_hh path is added after the MFMAs
            gs0 += dar; gs1 += daz; gs2 += dan;
            bsn += dgn;
            if (t > 0) {                                   // h_{-1} = 0: nothing to propagate, no weight gradient from step 0
                bf4v h, l;
                split4(dar, h, l); *(bf4v *)(&Gc[col * GP + oc]) = h; *(bf4v *)(&Gc[(16 + col) * GP + oc]) = l;
                split4(daz, h, l); *(bf4v *)(&Gc[col * GP + GH + oc]) = h; *(bf4v *)(&Gc[(16 + col) * GP + GH + oc]) = l;
                split4(dgn, h, l); *(bf4v *)(&Gc[col * GP + 2 * GH + oc]) = h; *(bf4v *)(&Gc[(16 + col) * GP + 2 * GH + oc]) = l;
                split4(hp, h, l); *(bf4v *)(&Hc[col * HP + oc]) = h; *(bf4v *)(&Hc[(16 + col) * HP + oc]) = l;
            }
            // the next step's operands are requested BEFORE this step's stores: memory operations retire in order, a wait for the loads would
            // otherwise wait for the stores' round trip as well
            if (t > 0 && !(a.ablate & 8)) load_step(t - 1);   // in flight during the products below
            if ((tmask >> t) & 1) {
                if (valid && !(a.ablate & 1)) {
                    float *o = a.dgi + (row * S + t) * G3 + oc;
                    *(f4v *)o = gs0; *(f4v *)(o + GH) = gs1; *(f4v *)(o + 2 * GH) = gs2;
                }
                gs0 = gs1 = gs2 = zero4;
            }
            // dW_hh of step t + 1 from ITS planes (the other buffer: nobody writes it before the next barrier).  It is not on the recurrence's
            // path, so it sits here, in front of the barrier: a wave that is early multiplies while the late ones finish their gate math
            if (pending && !(a.ablate & 2)) weight_grad_block(Gs + (pb ^ 1) * GBUF, Hs + (pb ^ 1) * HBUF, wave, lane, acc);
            pending = false;
            if (t == 0) break;
            __syncthreads();                               // ONE barrier per step: step t's planes are complete, step t + 1's have been read
            if (!(a.ablate & 4)) drec += gate_product<NL>(Gc, Wh, wl, wreg, col, grp);
            pending = true;
            pb ^= 1;
        }
    }
    store_weight_grad(a.dw_part, wave, lane, a.accumulate, acc);
    // column sums of dgh_n: the 16 lanes of a k group hold the 16 rows of the same 4 units
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = bsn[j];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
        bsn[j] = v;
    }
    if (col == 0) {
        f4v *o = (f4v *)(a.dbn_part + (int64_t)blockIdx.x * GH + oc);
        *o = a.accumulate ? *o + bsn : bsn;
    }
}

// ------------------------------------------------------------------------------------------------
// gru_bwd_in_kernel: for every fresh (tile, step) unit, t descending:
//   dx[row, t] = dgi[row, t]·W_ih     dW_ih += dgi[row, t]^T x[row, t]     db_i += dgi[row, t]
// ZOUT (CoreDiffusion: x = H, the aggregation's output rows relu(res_t)): instead of dx the kernel writes what agg_bwd_kernel gathers,
//   G_t = sum_{j >= t} dx_j [x_j > 0];   Z[row, t] = nested ? sum_{i >= t} G_i : G_t;   S0[row] = G_0
// (agg_bwd_prep_kernel's recurrences run in registers along the tile's steps; a step that repeats x has no entry of its row tagged with
// it, so nobody reads its Z).  Rows leave in matrix-row order (order[position]).
// PLANES: x arrives as the forward's fp16 planes + row scales (ctgcn_core_aggregate_split_f32), else as fp32 rows.
// ------------------------------------------------------------------------------------------------
struct BwdInArgs {
    int64_t rows;
    int32_t steps;
    const float *dgi;        // [rows, steps, 384], valid at fresh steps
    const float *wih;        // [384, 128]
    const uint32_t *tmask;
    const _Float16 *xp1, *xp2;   // PLANES: [rows * steps, 128] each
    const float *xps;            //         [rows * steps]
    const float *x;          // !PLANES: row-step e at x + e ldx
    int64_t ldx;
    float *dx;               // !ZOUT: [rows, steps, 128]
    float *Z, *S0;           // ZOUT: [n, steps, 128] / [n, 128] (null: no self loop) in matrix-row order
    const int32_t *order;    // ZOUT: matrix row of position p (null: p)
    int32_t nested;
    float *dw_part;          // [gridDim.x, 384, 128]
    float *dbi_part;         // [gridDim.x, 384]
    int32_t accumulate;
    int32_t ablate;          // diagnostic (CTGCN_BWD_ABLATE): 1 no stores, 2 no weight-gradient products, 4 no gate product, 8 loads of the first unit only
};

template <bool PLANES, bool ZOUT>
__global__ __launch_bounds__(512, 2) void gru_bwd_in_kernel(const BwdInArgs a)
{
    __shared__ __bf16 Gs[2 * 16 * GP];
    __shared__ __bf16 Xs[2 * 16 * HP];
    __shared__ bf8v Wl[8 * 12 * 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 15, grp = lane >> 4;
    const int oc = wave * 16 + 4 * grp;
    const int S = a.steps;

    bf8v Wh[12], wreg[1];
    load_wt_fragments<12>(a.wih, wave, col, grp, Wh, &Wl[(wave * 12) * 64 + lane], wreg);
    const bf8v *wl = &Wl[(wave * 12) * 64 + lane];
    __syncthreads();

    const f4v zero4 = {0.f, 0.f, 0.f, 0.f};
    f4v acc[3][8];
#pragma unroll
    for (int ut = 0; ut < 3; ++ut)
#pragma unroll
        for (int jt = 0; jt < 8; ++jt) acc[ut][jt] = zero4;
    f4v bsum = zero4;                                      // threads 0-95: column sums of dgi (4 gate columns each) over the block's units

    // staging roles: dgi 16 rows x 96 float4 -> three per thread (same columns for every unit: the bias sums stay per thread);
    // x 16 rows x 32 lanes x 4 values
    const int gr_[3] = {tid / 96, (tid + 512) / 96, (tid + 1024) / 96};
    const int gc_[3] = {(tid % 96) * 4, ((tid + 512) % 96) * 4, ((tid + 1024) % 96) * 4};
    const int xr = tid >> 5, xc = (tid & 31) * 4;
    const int64_t ntiles = (a.rows + 15) / 16;

    struct Unit { int64_t tile; int t; uint64_t mask; };
    auto mask_of = [&](int64_t tile) -> uint64_t { return tile < ntiles ? bwd_step_mask(a.tmask, tile, S) : ~0ull; };
    auto next_unit = [&](Unit &u) {                        // the unit after u in this block's order: fresh steps of a tile, descending
        do {
            if (--u.t < 0) { u.t = S - 1; u.tile += gridDim.x; u.mask = mask_of(u.tile); }
        } while (u.tile < ntiles && !((u.mask >> u.t) & 1));
    };
    f4v gv[3], xv;
    h4v xq1, xq2;
    float xsc = 0.f;
    int32_t orow_n = 0;                                    // ZOUT: matrix row of this lane's tile row, requested with the unit's operands
    auto load_unit = [&](const Unit u) {
        if (u.tile >= ntiles) return;
        const int64_t row0 = u.tile * 16;
        const int64_t lastrow = a.rows - 1;
        // (a load in the epilogue would be the youngest of the wave: waiting for it waits for every operand load of the NEXT unit too)
        if (ZOUT && a.order && u.t == bwd_top_step(u.mask, S)) orow_n = a.order[min(row0 + col, lastrow)];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int64_t r = min(row0 + gr_[i], lastrow);
            gv[i] = *(const f4v *)(a.dgi + (r * S + u.t) * G3 + gc_[i]);      // rows past the end are zeroed when they are STAGED: a select
        }                                                                        // here would wait for the load it follows
        const int64_t e = min(row0 + xr, lastrow) * S + u.t;
        if (PLANES) {
            xq1 = *(const h4v *)(a.xp1 + e * GH + xc);
            xq2 = *(const h4v *)(a.xp2 + e * GH + xc);
            xsc = a.xps[e];
        } else {
            xv = *(const f4v *)(a.x + e * a.ldx + xc);
        }
    };
    auto stage_unit = [&](const Unit u) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bf4v h, l;
            split4(u.tile * 16 + gr_[i] < a.rows ? gv[i] : zero4, h, l);
            *(bf4v *)(&Gs[gr_[i] * GP + gc_[i]]) = h;
            *(bf4v *)(&Gs[(16 + gr_[i]) * GP + gc_[i]]) = l;
        }
        f4v x;
        if (PLANES) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = ((float)xq1[j] + (float)xq2[j]) * xsc;
        } else {
            x = xv;
        }
        bf4v h, l;
        split4(x, h, l);
        *(bf4v *)(&Xs[xr * HP + xc]) = h;
        *(bf4v *)(&Xs[(16 + xr) * HP + xc]) = l;
    };

    Unit cur{(int64_t)blockIdx.x, S, mask_of(blockIdx.x)};
    do { --cur.t; } while (cur.t > 0 && !((cur.mask >> cur.t) & 1));      // the first tile's last fresh step (bit 0 is always set)
    if (cur.tile < ntiles) load_unit(cur);
    f4v Gr = zero4, Zr = zero4;
    int32_t orow_c = 0;
    while (cur.tile < ntiles) {
        if (ZOUT && a.order) {                             // the first unit of a tile brings the tile's row map
            const int top = bwd_top_step(cur.mask, S);
            if (cur.t == top) orow_c = orow_n;
        }
        stage_unit(cur);
        Unit nxt = cur;
        next_unit(nxt);
        if (!(a.ablate & 8)) load_unit(nxt);               // in flight during the products
        __syncthreads();
        f4v dxv = zero4;
        if (!(a.ablate & 4)) dxv = gate_product<12>(Gs, Wh, wl, wreg, col, grp);
        if (!(a.ablate & 2)) weight_grad_block(Gs, Xs, wave, lane, acc);
        if (tid < 96) {
            // d b_ih: column sums of the unit's dgi rows, read back from the planes (hi + lo = the value to 2^-17: what the products see);
            // 96 threads x 4 columns — per-thread sums over the staging registers cost 12 registers on every thread and spilled
#pragma unroll 4
            for (int r = 0; r < 16; ++r) {
                const bf4v h = *(const bf4v *)(&Gs[r * GP + 4 * tid]), l = *(const bf4v *)(&Gs[(16 + r) * GP + 4 * tid]);
#pragma unroll
                for (int j = 0; j < 4; ++j) bsum[j] += (float)h[j] + (float)l[j];
            }
        }
        const int64_t row0 = cur.tile * 16;
        const bool valid = row0 + col < a.rows && !(a.ablate & 1);
        if (ZOUT) {
            const bf4v xh = *(const bf4v *)(&Xs[col * HP + oc]);
            f4v g = dxv;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = (float)xh[j] > 0.f ? g[j] : 0.f;
            Gr += g;
            Zr += Gr;
            if (valid) {
                const int64_t orow = a.order ? (int64_t)orow_c : row0 + col;
                *(f4v *)(a.Z + (orow * S + cur.t) * GH + oc) = a.nested ? Zr : Gr;
                if (cur.t == 0 && a.S0) *(f4v *)(a.S0 + orow * GH + oc) = Gr;
            }
            if (cur.t == 0) { Gr = zero4; Zr = zero4; }   // step 0 is a tile's last unit
        } else {
            if (valid) *(f4v *)(a.dx + ((row0 + col) * S + cur.t) * GH + oc) = dxv;
        }
        __syncthreads();                                   // the planes are rewritten by the next unit
        cur = nxt;
    }
    store_weight_grad(a.dw_part, wave, lane, a.accumulate, acc);
    if (tid < 96) {
        f4v *o = (f4v *)(a.dbi_part + (int64_t)blockIdx.x * G3 + tid * 4);
        *o = a.accumulate ? *o + bsum : bsum;
    }
}

int device_cus(int *cus)
{
    int dev = 0;
    *cus = 256;
    BWD_TRY(hipGetDevice(&dev));
    BWD_TRY(hipDeviceGetAttribute(cus, hipDeviceAttributeMultiprocessorCount, dev));
    *cus = ctgcn_persistent_cus_(*cus);
    return CTGCN_OK;
}
bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int ablate_mask() { static const int m = [] { const char *e = getenv("CTGCN_BWD_ABLATE"); return e ? atoi(e) : 0; }(); return m; }

}  // namespace

extern "C" {

int32_t ctgcn_gru_bwd_blocks(int64_t rows)
{
    int cus = 256;
    if (device_cus(&cus) != CTGCN_OK) cus = 256;
    const int64_t ntiles = (rows + 15) / 16;
    return (int32_t)(ntiles < cus ? (ntiles > 0 ? ntiles : 1) : cus);
}

int ctgcn_gru_bwd_rec_f32(int64_t rows, int32_t steps, int32_t hidden, const float *gates, int32_t gate_count, const float *h_seq, const float *dh_sum,
                          const float *dh_seq, const float *w_hh, const uint32_t *tile_mask, float *d_gi, float *dw_partial,
                          float *dbn_partial, int32_t n_partial, int32_t accumulate, void *stream)
{
    if (hidden != GH) return ctgcn_set_error_(CTGCN_E_UNSUPPORTED, "gru_bwd_rec: only hidden = 128 is built");
    if (rows < 0 || steps < 1 || steps > 64 || (gate_count != 3 && gate_count != 4)) return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_rec: bad sizes (1 <= steps <= 64, gate_count 3 or 4)");
    if (!gates || !h_seq || !w_hh || !d_gi || !dw_partial || !dbn_partial || ((dh_sum == nullptr) == (dh_seq == nullptr)))
        return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_rec: null pointer (exactly one of dh_sum / dh_seq)");
    if (!aligned16(gates) || !aligned16(h_seq) || !aligned16(dh_sum) || !aligned16(dh_seq) || !aligned16(d_gi) || !aligned16(dw_partial) || !aligned16(dbn_partial))
        return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_rec: buffers must be 16-byte aligned");
    const int32_t blocks = ctgcn_gru_bwd_blocks(rows);
    if (n_partial < blocks) return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_rec: n_partial < ctgcn_gru_bwd_blocks(rows)");
    if (rows == 0) return CTGCN_OK;
    BwdRecArgs a{};
    a.rows = rows; a.steps = steps; a.gates = gates; a.gates3 = gate_count == 3 ? 1 : 0; a.hseq = h_seq; a.dh = dh_sum ? dh_sum : dh_seq; a.whh = w_hh; a.tmask = tile_mask;
    a.dgi = d_gi; a.dw_part = dw_partial; a.dbn_part = dbn_partial; a.accumulate = accumulate ? 1 : 0; a.ablate = ablate_mask();
    if (steps > 32) {          // two mask words per tile (or no plan: all ones over 64 bits)
        if (dh_sum) hipLaunchKernelGGL((gru_bwd_rec_kernel<true, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gru_bwd_rec_kernel<false, true>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    } else if (dh_sum) hipLaunchKernelGGL(gru_bwd_rec_kernel<true>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gru_bwd_rec_kernel<false>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, a);
    BWD_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_gru_bwd_in_f32(int64_t rows, int32_t steps, int32_t hidden, const float *d_gi, const float *w_ih, const uint32_t *tile_mask,
                         const void *x_planes, int64_t plane_rows, int64_t first_row, const float *x, int64_t ldx, float *dx, float *Z, float *S0,
                         const int32_t *row_order, int32_t nested, float *dw_partial, float *dbi_partial, int32_t n_partial,
                         int32_t accumulate, void *stream)
{
    if (hidden != GH) return ctgcn_set_error_(CTGCN_E_UNSUPPORTED, "gru_bwd_in: only d_in = hidden = 128 is built");
    if (rows < 0 || steps < 1 || steps > 64) return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_in: bad sizes (1 <= steps <= 64)");
    if (!d_gi || !w_ih || !dw_partial || !dbi_partial || ((x_planes == nullptr) == (x == nullptr)) || ((dx == nullptr) == (Z == nullptr)))
        return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_in: null pointer (exactly one of x_planes / x and of dx / Z)");
    if (!aligned16(d_gi) || !aligned16(x) || !aligned16(dx) || !aligned16(Z) || !aligned16(S0) || !aligned16(dw_partial) || !aligned16(dbi_partial) ||
        (reinterpret_cast<uintptr_t>(x_planes) & 7u) || (x && (ldx < GH || (ldx & 3))))
        return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_in: buffers must be 16-byte aligned (planes 8), ldx a multiple of 4 and >= 128");
    if (x_planes && (first_row < 0 || (first_row & 15) || plane_rows < (first_row + rows) * steps))
        return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_in: first_row must be a multiple of 16 and (first_row + rows) * steps <= plane_rows");
    const int32_t blocks = ctgcn_gru_bwd_blocks(rows);
    if (n_partial < blocks) return ctgcn_set_error_(CTGCN_E_INVALID, "gru_bwd_in: n_partial < ctgcn_gru_bwd_blocks(rows)");
    if (rows == 0) return CTGCN_OK;
    BwdInArgs a{};
    a.rows = rows; a.steps = steps; a.dgi = d_gi; a.wih = w_ih; a.tmask = tile_mask;
    if (x_planes) {          // the layout ctgcn_core_aggregate_split_f32 writes for d = 128: plane 1, plane 2, row scales over plane_rows rows
        // this call takes the sequences [first_row, first_row + rows) of it; tile_mask / row_order already point at first_row
        a.xp1 = (const _Float16 *)x_planes + (size_t)first_row * steps * GH;
        a.xp2 = (const _Float16 *)x_planes + ((size_t)plane_rows + (size_t)first_row * steps) * GH;
        a.xps = (const float *)((const _Float16 *)x_planes + 2 * (size_t)plane_rows * GH) + (size_t)first_row * steps;
    }
    a.x = x; a.ldx = ldx; a.dx = dx; a.Z = Z; a.S0 = S0; a.order = row_order; a.nested = nested ? 1 : 0;
    a.dw_part = dw_partial; a.dbi_part = dbi_partial; a.accumulate = accumulate ? 1 : 0; a.ablate = ablate_mask();
    hipStream_t st = (hipStream_t)stream;
    if (x_planes && Z) hipLaunchKernelGGL((gru_bwd_in_kernel<true, true>), dim3((unsigned)blocks), dim3(512), 0, st, a);
    else if (x_planes) hipLaunchKernelGGL((gru_bwd_in_kernel<true, false>), dim3((unsigned)blocks), dim3(512), 0, st, a);
    else if (Z) hipLaunchKernelGGL((gru_bwd_in_kernel<false, true>), dim3((unsigned)blocks), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((gru_bwd_in_kernel<false, false>), dim3((unsigned)blocks), dim3(512), 0, st, a);
    BWD_TRY(hipGetLastError());
    return CTGCN_OK;
}

}  // extern "C"
