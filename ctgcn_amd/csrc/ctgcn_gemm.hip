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
// Three kernels:
//   split_rows_h2_kernel   fp32 rows -> per-row scale + the two fp16 planes [rows, Kp] (Kp = K rounded up to 64, zero padded): the X side.
//   pack_weight_h2_kernel  fp32 weight rows -> the same split, stored in MFMA-fragment order (the W side; once per weight version).
//   gemm_h2_panel_kernel   (round 5) persistent blocks of 8 waves; a block owns 128-row PANELS of X over the FULL width N <= 512, so X's
//                          planes leave HBM exactly once; X is staged through LDS by direct-to-LDS loads of whole 128-byte lines (k stages
//                          of 64), W fragments go from L2 straight into registers (each wave owns its own N / 8 columns: no LDS, no
//                          sharing), the next panel's operands are in flight while a panel's epilogue stores.  See the kernel's comment.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

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

size_t align_up(size_t x, size_t al) { return (x + al - 1) / al * al; }

constexpr int PBM = 128;          // rows of a panel
constexpr int PSK = 64;           // k per stage: one 128-byte line of a plane row
constexpr int PCHUNK = 512;       // widest N one launch covers (8 waves x 4 column tiles of 16)

// ------------------------------------------------------------------------------------------------ W side
// Packed weight operand: per chunk of <= 512 output columns  [column tile ct of 16][k slab of 32][plane][lane][8 halfs], i.e. the 1 KB
// one global_load_dwordx4 of a wave fetches is exactly the v_mfma_f32_16x16x32_f16 operand of (ct, slab, plane): lane l holds
// W[ct 16 + (l & 15)][slab 32 + 8 (l >> 4) .. + 7].  A chunk's column tiles are padded to 8 NT (NT tiles per wave), k to Kp, with zeros.
// Behind the fragments of all chunks: one power-of-two scale per (padded) column.  Same split as split_rows_h2_kernel.
__host__ __device__ inline int chunk_nt(int cols) { return ((cols + 15) / 16 + 7) / 8; }          // column tiles per wave for a chunk of `cols` columns

struct PackGeom {
    int32_t chunks, kp;
    size_t frag_halfs;       // halfs of all chunks' fragments
    int32_t npad;            // padded columns of all chunks
};
inline PackGeom pack_geom(int32_t n_out, int32_t k)
{
    PackGeom g{};
    g.kp = (int32_t)align_up((size_t)k, PSK);
    g.chunks = (n_out + PCHUNK - 1) / PCHUNK;
    for (int c = 0; c < g.chunks; ++c) {
        const int cols = n_out - c * PCHUNK < PCHUNK ? n_out - c * PCHUNK : PCHUNK;
        g.npad += chunk_nt(cols) * 128;
    }
    g.frag_halfs = (size_t)g.npad * g.kp * 2;
    return g;
}

// one wave per (padded) weight row
__global__ __launch_bounds__(256) void pack_weight_h2_kernel(int32_t n_out, int32_t npad, int32_t K, int32_t Kp, const float *__restrict__ w, int64_t ldw,
                                                              _Float16 *__restrict__ packed, float *__restrict__ sb)
{
    const int lane = threadIdx.x & 63;
    const int prow = blockIdx.x * 4 + (threadIdx.x >> 6);          // padded row index over all chunks
    if (prow >= npad) return;
    // padded row -> (chunk, column inside the chunk): every chunk but the last is exactly PCHUNK wide (NT = 4, no padding)
    const int full = n_out / PCHUNK;                                // chunks of exactly 512 columns
    const int chunk = prow / PCHUNK < full ? prow / PCHUNK : full;
    const int cin = prow - chunk * PCHUNK;                          // column inside the chunk (incl. its padding)
    const int col = chunk * PCHUNK + cin;
    const int cols = n_out - chunk * PCHUNK < PCHUNK ? n_out - chunk * PCHUNK : PCHUNK;
    const bool live = cin < cols;
    const float *src = w + (int64_t)col * ldw;
    float m = 0.f;
    if (live) for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(src[k]));
#pragma unroll
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s, inv;
    h2_scale(m, s, inv);
    if (lane == 0) sb[prow] = live ? s : 1.f;
    const int KS = Kp / 32;
    _Float16 *base = packed + (size_t)chunk * PCHUNK * Kp * 2;      // fragments of the chunks before this one (each 512 x Kp x 2 planes)
    const int ct = cin >> 4, i = cin & 15;
    for (int g = lane; g < Kp / 8; g += 64) {
        const int k0 = g * 8;
        h8v hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xs = (live && k0 + e < K) ? src[k0 + e] * inv : 0.f;
            hi[e] = (_Float16)xs;
            lo[e] = (_Float16)(xs - (float)hi[e]);
        }
        _Float16 *dst = base + (((size_t)ct * KS + (k0 >> 5)) * 2 * 64 + (i + 16 * ((k0 & 31) >> 3))) * 8;
        *(h8v *)dst = hi;
        *(h8v *)(dst + 512) = lo;
    }
}

// ------------------------------------------------------------------------------------------------ the GEMM
struct PanelArgs {
    int64_t M;
    int32_t N, Kp;                           // columns of this chunk, padded k
    const _Float16 *a1, *a2;                 // X planes [M, Kp]
    const float *sa;                         // X row scales
    const _Float16 *bp;                      // this chunk's packed fragments
    const float *sb, *bias;                  // this chunk's column scales (padded), bias (N entries) or null
    int32_t act;                             // 0 none, 1 SELU (layers.py:103-104)
    int32_t vec;                             // y rows take 16-byte stores (ldy % 4 == 0, 16-byte aligned base)
    float *y;                                // first column of this chunk
    int64_t ldy;
    int64_t mtiles;
};

#ifndef CTGCN_GEMM_ABLATE
#define CTGCN_GEMM_ABLATE 0          // diagnostic builds (WRONG results): 1 no W-fragment loads in the loop, 2 no X staging in the loop, 4 no MFMA, 5 no epilogue stores
#endif
#ifndef CTGCN_GEMM_NT_LOADS
#define CTGCN_GEMM_NT_LOADS 1        // X planes are streamed once: non-temporal direct-to-LDS loads keep them from evicting W from the XCD's L2
#endif

// gemm_h2_panel_kernel<NT>: Y[M, N] = X W^T for N <= 128 NT.  Persistent blocks of 8 waves (two per SIMD), one block per CU.
//   * A block works on PANELS of 128 rows of X over the full width: X's planes are read from HBM exactly once (round 4's 128 x 128 tiles
//     re-read them N / 128 times through L2, in 64-byte pieces).  k advances in STAGES of 64: per row and plane one whole 128-byte line,
//     fetched by global_load_lds_dwordx4 into a three-slot LDS ring (32 KB per stage, requested two stages ahead; XOR swizzle applied on the global side so that the
//     fragment reads are conflict-free), no registers, no ds_write.
//   * Wave w owns output columns [16 NT w, 16 NT (w + 1)): its W fragments come from the packed layout (above) with one coalesced 1 KB load
//     per (column tile, k slab, plane) STRAIGHT INTO REGISTERS — W is never in LDS, nobody shares it, no barrier guards it.  Every wave reads
//     all 128 X rows of the stage from LDS (8 x 32 KB of ds_read_b128 per stage: 22 % of the LDS read rate at NT = 3).
//   * Products: v_mfma_f32_16x16x32_f16 with W as the first operand, so a lane ends up with FOUR CONSECUTIVE COLUMNS of one output row
//     (16-byte stores); per (row tile, column tile, k slab) three MFMAs into one accumulator, small terms first (x1 w2, x2 w1, x1 w1).
//   * One wait + one barrier per stage: `s_waitcnt vmcnt(0)` at the top of a stage retires, youngest first, the W fragments of the stage's
//     first slab (requested half a stage ago), the X lines of this stage (requested a stage ago) — memory operations retire in order, so
//     one counter serves both streams.  Epilogue stores of a finished panel are issued at the start of the NEXT panel's first stage, behind
//     that stage's operand requests: they drain under a whole stage of MFMAs, and the next panel's operands were already in flight.
//   * Registers (NT = 3): 96 accumulators + 48 W fragments (two slabs) + 32 X fragments (two row-tile pairs) = 176 + addresses.
template <int NT>
__global__ __launch_bounds__(512, 2) void gemm_h2_panel_kernel(const PanelArgs a)
{
    constexpr int PLANE = PBM * PSK;                      // halfs of one plane of a stage (16 KB)
    __shared__ __attribute__((aligned(1024))) _Float16 As[3][2 * PLANE];      // [slot][plane][row][64 k], 16-byte segment g of row r at g ^ (r & 7)
    __shared__ __attribute__((aligned(16))) float s_sb[NT * 128];
    __shared__ __attribute__((aligned(16))) float s_bias[NT * 128];
    __shared__ float s_sa[2][PBM];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;        // wave: in an SGPR
    const int nks = a.Kp / PSK, KS = a.Kp / 32;
    const int64_t stride = gridDim.x;
    if ((int64_t)blockIdx.x >= a.mtiles) return;
    const int64_t npan = (a.mtiles - blockIdx.x + stride - 1) / stride;
    const int64_t total = npan * nks;

    for (int i = tid; i < NT * 128; i += 512) {
        s_sb[i] = a.sb[i];
        s_bias[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    }

    // X staging: a wave instruction fills 8 rows x 128 B of one plane linearly (lane l -> row l >> 3, stored segment l & 7), so the lane fetches
    // the segment that belongs there: (l & 7) ^ (row & 7).  32 instructions per stage, four per wave: waves 0-3 plane 1, waves 4-7 plane 2.
    const _Float16 *aplane = (wave < 4) ? a.a1 : a.a2;
    const int rg0 = (wave & 3) * 32;                      // first of this wave's 32 rows
    const int lrow = lane >> 3;
    const int gseg = ((lane & 7) ^ lrow) * 8;
    // As an INLINE-ASM instruction: behind the compiler's own LDS-DMA intrinsic every ds_read of the staging buffer waits for vmcnt(0) — the
    // compiler cannot tell which slot a request fills — i.e. for the requests issued a moment earlier (seen in the ISA: prefetch distance 0).
    // Completion is this kernel's business: `s_waitcnt vmcnt` + s_barrier below.  M0 = LDS base of the 1 KB the instruction fills.
    const uint32_t ldst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) _Float16 *)&As[0][(wave >> 2) * PLANE + rg0 * PSK];
    auto dma = [&](int64_t pn, int ksn, int slot) __attribute__((always_inline)) {
        if (CTGCN_GEMM_ABLATE == 2 && (pn != (int64_t)blockIdx.x || ksn > 1)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = min(pn * PBM + rg0 + i * 8 + lrow, a.M - 1);
            const _Float16 *src = aplane + row * a.Kp + ksn * PSK + gseg;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ldst + (uint32_t)(slot * 2 * PLANE + i * 8 * PSK) * 2u);
#if CTGCN_GEMM_NT_LOADS
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src), "s"(dst) : "memory");
#else
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
#endif
        }
    };

    // W fragments: column tile (wave NT + j), slab s, plane p -> 512 halfs at ((j KS + s) 2 + p) 512 from the wave's base.  Scalar base +
    // one 32-bit lane offset: the address costs no vector registers (eight 64-bit addresses would be 16 of them)
    const _Float16 *const bwave = a.bp + (size_t)wave * NT * KS * 1024;
    const uint32_t blane = lane * 16;                     // bytes
    h8v fb[2][NT][2];
    auto loadB = [&](int buf, int s) __attribute__((always_inline)) {
        if (CTGCN_GEMM_ABLATE == 1 && s > 1) return;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) fb[buf][j][p] = *(const h8v *)((const char *)(bwave + ((size_t)(j * KS + s) * 2 + p) * 512) + blane);
    };

    // X fragments: row tile r, slab sl: lane l reads row r 16 + (l & 15), k = sl 32 + 8 (l >> 4) .. + 7  (segment sl 4 + (l >> 4))
    const int arow = lane & 15;
    int aoff[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) aoff[sl] = arow * PSK + (((sl * 4 + (lane >> 4)) ^ (lane & 7)) * 8);

    f4v acc[8][NT];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[r][j] = f4v{0.f, 0.f, 0.f, 0.f};

    // all MFMAs of one k slab: row tiles in groups of RG (2; 1 at NT = 4, where 128 accumulators + 64 W fragment registers leave no room for
    // four X fragments more), the next group's fragments requested before this group's 3 RG NT products.  An accumulator is written by every
    // RG NT-th MFMA: >= 3 issue slots apart
    constexpr int RG = NT >= 4 ? 1 : 2, NG = 8 / RG;
    auto slab = [&](int slot, int sl, int buf) __attribute__((always_inline)) {
        const _Float16 *sbase = &As[slot][aoff[sl]];
        h8v xa[2][RG][2];                                 // [parity][row tile of the group][plane]
#pragma unroll
        for (int rr = 0; rr < RG; ++rr)
#pragma unroll
            for (int p = 0; p < 2; ++p) xa[0][rr][p] = *(const h8v *)(sbase + p * PLANE + rr * 16 * PSK);
#pragma unroll
        for (int rp = 0; rp < NG; ++rp) {
            const int cur = rp & 1;
            if (rp + 1 < NG) {
#pragma unroll
                for (int rr = 0; rr < RG; ++rr)
#pragma unroll
                    for (int p = 0; p < 2; ++p) xa[cur ^ 1][rr][p] = *(const h8v *)(sbase + p * PLANE + ((rp + 1) * RG + rr) * 16 * PSK);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int rr = 0; rr < RG; ++rr)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (CTGCN_GEMM_ABLATE != 4)
                            acc[rp * RG + rr][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[buf][j][term == 0 ? 1 : 0], xa[cur][rr][term == 1 ? 1 : 0], acc[rp * RG + rr][j], 0, 0, 0);
                        else asm volatile("" :: "v"(fb[buf][j][term == 0 ? 1 : 0]), "v"(xa[cur][rr][term == 1 ? 1 : 0]));
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // y = act(acc sa[m] sb[n] + bias[n]): lane l holds columns 4 (l >> 4) .. + 3 of row (l & 15) of every 16 x 16 tile.  A panel that lies
    // inside the matrix (rows) whose wave's columns all exist takes the straight-line path — 8 NT 16-byte stores back to back; with a test
    // around every store the compiler branches around each one and waits for vmcnt(0) in front of it (stores count in vmcnt on gfx9).
    auto epilogue_as = [&](int64_t pn, int par, auto selu, auto whole) __attribute__((always_inline)) {
        const int64_t m0 = pn * PBM;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int64_t m = m0 + r * 16 + arow;
            const float s = s_sa[par][r * 16 + arow];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (wave * NT + j) * 16 + 4 * (lane >> 4);
                const f4v sbv = *(const f4v *)&s_sb[n], bv = *(const f4v *)&s_bias[n];
                f4v o;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float t = fmaf(acc[r][j][v], s * sbv[v], bv[v]);
                    if (decltype(selu)::value) {          // torch's SELU, both sides evaluated (no branch): expm1 keeps the accuracy near 0
                        const float neg = (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f) * expm1f(fminf(t, 0.f));
                        o[v] = t > 0.f ? 1.0507009873554804934193349852946f * t : neg;
                    } else o[v] = t;
                }
                acc[r][j] = f4v{0.f, 0.f, 0.f, 0.f};
                if (CTGCN_GEMM_ABLATE == 5) { asm volatile("" :: "v"(o)); continue; }
                float *dst = a.y + m * a.ldy + n;
                if (decltype(whole)::value) *(f4v *)dst = o;
                else if (m < a.M) {
                    if (a.vec && n + 3 < a.N) *(f4v *)dst = o;
                    else {
#pragma unroll
                        for (int v = 0; v < 4; ++v) if (n + v < a.N) dst[v] = o[v];
                    }
                }
            }
        }
    };
    const bool cols_whole = a.vec && (wave * NT + NT) * 16 <= a.N;
    auto epilogue = [&](int64_t pn, int par) __attribute__((always_inline)) {
        const bool whole = cols_whole && (pn + 1) * PBM <= a.M;
        if (a.act == 1) {
            if (whole) epilogue_as(pn, par, std::true_type{}, std::true_type{});
            else epilogue_as(pn, par, std::true_type{}, std::false_type{});
        } else {
            if (whole) epilogue_as(pn, par, std::false_type{}, std::true_type{});
            else epilogue_as(pn, par, std::false_type{}, std::false_type{});
        }
    };

    int64_t pan_c = blockIdx.x, pan_2 = blockIdx.x;       // the stage being multiplied: (panel, k stage); the stage requested two ahead
    int ks_c = 0, ks_2 = 0, par = 0, slot = 0;
    float sa_r = 0.f;
    bool sa_pending = false;
    int sa_par = 0;
    auto advance = [&](int64_t &pn, int &ks) __attribute__((always_inline)) { if (++ks == nks) { ks = 0; pn += stride; } };
    dma(pan_2, ks_2, 0);
    advance(pan_2, ks_2);
    dma(pan_2, ks_2, 1);
    advance(pan_2, ks_2);
    loadB(0, 0);
    for (int64_t it = 0; it < total; ++it) {
        int64_t pan_n = pan_c;
        int ks_n = ks_c;
        advance(pan_n, ks_n);
        // everything requested so far has landed: this wave's share of this stage's X lines (and the next one's), the W fragments of slab 0, the scales
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (sa_pending) {
            if (tid < PBM) s_sa[sa_par][tid] = sa_r;
            sa_pending = false;
        }
        __syncthreads();                                  // the stage is complete for every wave; the slot of stage it - 1 has been read by every wave
        // everything conditional comes BEFORE this stage's operand requests: behind a branch the compiler cannot count the outstanding
        // loads and waits for vmcnt(0) at the next use of a W fragment — i.e. for the requests issued a moment ago (measured in the ISA)
        if (ks_c == 0) {
            if (tid < PBM) sa_r = a.sa[min(pan_c * PBM + tid, a.M - 1)];
            sa_pending = true;
            sa_par = par;
            if (it > 0) epilogue(pan_c - stride, par ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        loadB(1, ks_c * 2 + 1);
        dma(pan_2, ks_2, slot == 0 ? 2 : slot - 1);       // stage it + 2 into the slot of stage it - 1; never conditional (past the end: a valid address, a dead slot)
        advance(pan_2, ks_2);
        __builtin_amdgcn_sched_barrier(0);
        slab(slot, 0, 0);
        loadB(0, ks_n * 2);                               // the next stage's first slab (past the end: a valid, unused address)
        __builtin_amdgcn_sched_barrier(0);
        slab(slot, 1, 1);                                 // (the compiler's wait for these fragments also retires the X requests above: in order)
        if (ks_n == 0) par ^= 1;
        pan_c = pan_n;
        ks_c = ks_n;
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sa_pending && tid < PBM) s_sa[sa_par][tid] = sa_r;
    __syncthreads();
    epilogue(pan_c - stride, par ^ 1);
}

int device_cus()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

}  // namespace

extern "C" {

size_t ctgcn_split_planes_bytes(int64_t rows, int32_t k)
{
    if (rows < 0 || k < 1) return 0;
    const size_t kp = align_up((size_t)k, PSK);
    return align_up((size_t)rows * kp * 4 + (size_t)rows * 4, 256);
}

size_t ctgcn_pack_weight_bytes(int32_t n_out, int32_t k)
{
    if (n_out < 1 || k < 1) return 0;
    const PackGeom g = pack_geom(n_out, k);
    return align_up(g.frag_halfs * 2 + (size_t)g.npad * 4, 256);
}

size_t ctgcn_linear_workspace_bytes(int64_t rows, int32_t n_out, int32_t k)
{
    if (rows < 0 || n_out < 1 || k < 1) return 0;
    return ctgcn_split_planes_bytes(rows, k) + ctgcn_pack_weight_bytes(n_out, k) + 256;
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

int ctgcn_split_rows_f32(int64_t rows, int32_t k, const float *x, int64_t ldx, void *planes, size_t planes_bytes, void *stream)
{
    if (rows < 0 || k < 1 || ldx < k) return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows: bad sizes");
    if (rows == 0) return CTGCN_OK;
    if (!x || !planes || (reinterpret_cast<uintptr_t>(x) & 3u) || (reinterpret_cast<uintptr_t>(planes) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "split_rows: x 4-byte aligned, planes 256-byte aligned");
    if (planes_bytes < ctgcn_split_planes_bytes(rows, k)) return ctgcn_set_error_(CTGCN_E_WORKSPACE, "split_rows: planes buffer too small (ctgcn_split_planes_bytes)");
    const int32_t kp = (int32_t)align_up((size_t)k, PSK);
    _Float16 *p1 = (_Float16 *)planes, *p2 = p1 + (size_t)rows * kp;
    float *sc = (float *)(p2 + (size_t)rows * kp);
    launch_split(rows, k, kp, x, ldx, p1, p2, sc, nullptr, 1, 1.f, (hipStream_t)stream);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_pack_weight_f32(int32_t n_out, int32_t k, const float *w, int64_t ldw, void *packed, size_t packed_bytes, void *stream)
{
    if (n_out < 1 || k < 1 || ldw < k) return ctgcn_set_error_(CTGCN_E_INVALID, "pack_weight: bad sizes");
    if (!w || !packed || (reinterpret_cast<uintptr_t>(w) & 3u) || (reinterpret_cast<uintptr_t>(packed) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "pack_weight: w 4-byte aligned, packed 256-byte aligned");
    if (packed_bytes < ctgcn_pack_weight_bytes(n_out, k)) return ctgcn_set_error_(CTGCN_E_WORKSPACE, "pack_weight: buffer too small (ctgcn_pack_weight_bytes)");
    const PackGeom g = pack_geom(n_out, k);
    _Float16 *frags = (_Float16 *)packed;
    float *sb = (float *)(frags + g.frag_halfs);
    hipLaunchKernelGGL(pack_weight_h2_kernel, dim3((unsigned)((g.npad + 3) / 4)), dim3(256), 0, (hipStream_t)stream, n_out, g.npad, k, g.kp, w, ldw, frags, sb);
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_linear_packed_f32(int64_t rows, int32_t n_out, int32_t k, const void *x_planes, const void *w_packed, const float *bias, int32_t activation,
                            float *y, int64_t ldy, void *stream)
{
    if (rows < 0 || n_out < 1 || k < 1 || ldy < n_out) return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed: bad sizes");
    if (activation != CTGCN_ACT_NONE && activation != CTGCN_ACT_SELU) return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed: unknown activation");
    if (rows == 0) return CTGCN_OK;
    if (!x_planes || !w_packed || !y || (reinterpret_cast<uintptr_t>(x_planes) & 255u) || (reinterpret_cast<uintptr_t>(w_packed) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed: null or misaligned (256 bytes) operand buffers");
    const PackGeom g = pack_geom(n_out, k);
    PanelArgs a{};
    a.M = rows; a.Kp = g.kp;
    a.a1 = (const _Float16 *)x_planes; a.a2 = a.a1 + (size_t)rows * g.kp; a.sa = (const float *)(a.a2 + (size_t)rows * g.kp);
    a.act = activation; a.ldy = ldy;
    a.mtiles = (rows + PBM - 1) / PBM;
    const _Float16 *frags = (const _Float16 *)w_packed;
    const float *sb = (const float *)(frags + g.frag_halfs);
    const int64_t blocks = a.mtiles < device_cus() ? a.mtiles : device_cus();
    size_t frag_off = 0;
    int32_t pad_off = 0;
    for (int c = 0; c < g.chunks; ++c) {
        const int cols = n_out - c * PCHUNK < PCHUNK ? n_out - c * PCHUNK : PCHUNK;
        const int nt = chunk_nt(cols);
        a.N = cols; a.bp = frags + frag_off; a.sb = sb + pad_off; a.bias = bias ? bias + c * PCHUNK : nullptr;
        a.y = y + c * PCHUNK;
        a.vec = (!(ldy & 3) && !(reinterpret_cast<uintptr_t>(a.y) & 15u)) ? 1 : 0;
        const dim3 grid((unsigned)blocks), blk(512);
        switch (nt) {
        case 1: hipLaunchKernelGGL(gemm_h2_panel_kernel<1>, grid, blk, 0, (hipStream_t)stream, a); break;
        case 2: hipLaunchKernelGGL(gemm_h2_panel_kernel<2>, grid, blk, 0, (hipStream_t)stream, a); break;
        case 3: hipLaunchKernelGGL(gemm_h2_panel_kernel<3>, grid, blk, 0, (hipStream_t)stream, a); break;
        default: hipLaunchKernelGGL(gemm_h2_panel_kernel<4>, grid, blk, 0, (hipStream_t)stream, a); break;
        }
        frag_off += (size_t)nt * 128 * g.kp * 2;
        pad_off += nt * 128;
    }
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

// x == nullptr: the X planes and scales are already at the head of the workspace (ctgcn_core_aggregate_split_f32 wrote them)
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
        return ctgcn_set_error_(CTGCN_E_WORKSPACE, "linear: workspace too small (ctgcn_linear_workspace_bytes)");
    const size_t xbytes = ctgcn_split_planes_bytes(rows, k);
    char *ws = (char *)workspace;
    if (x) {
        if (int rc = ctgcn_split_rows_f32(rows, k, x, ldx, ws, xbytes, stream)) return rc;
    }
    if (int rc = ctgcn_pack_weight_f32(n_out, k, w, ldw, ws + xbytes, ctgcn_pack_weight_bytes(n_out, k), stream)) return rc;
    return ctgcn_linear_packed_f32(rows, n_out, k, ws, ws + xbytes, bias, act, y, ldy, stream);
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
