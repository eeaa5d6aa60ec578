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
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/ctgcn_hip.h"
#include "ctgcn_jitter.h"
#include "ctgcn_table.h"           // descriptor tables of the grouped launches: kernel-argument upload + host shadow          // diagnostic builds (-DCTGCN_JITTER): delays around every barrier; nothing in the product

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
struct PanelGroup { const _Float16 *bp; const float *sb, *bias; };
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
    // grouped launch (round 5: the projections of a small window's snapshots in ONE launch): the X planes, the row scales and Y hold the rows of all
    // groups one after the other (every group padded to whole panels), W / column scales / bias are per group: panel -> group, group -> operands
    const int32_t *panel_group;              // [mtiles] or null (one group: bp / sb / bias above)
    const struct PanelGroup *groups;
    // chained launch (CHAIN instantiation): the output leaves as the NEXT layer's X operand — per-row scale + two fp16 planes [M, okp],
    // okp = N rounded up to 64 — instead of fp32 rows: what split_rows_h2_kernel would make of y, bit for bit
    _Float16 *o1, *o2;
    float *osc;
    int32_t okp;
#ifdef CTGCN_GEMM_TIMELINE
    unsigned long long *timeline;            // diagnostic build: [block][wave 0 / 7][stage < 48][6] s_memtime stamps, see tools/gemm_timeline.py
#endif
};

#ifndef CTGCN_GEMM_ABLATE
#define CTGCN_GEMM_ABLATE 0          // diagnostic builds (WRONG results), a bit mask: 1 no W-fragment loads in the loop, 2 no X staging in the loop, 4 no MFMA, 8 no epilogue stores
#endif
#ifndef CTGCN_GEMM_NT_LOADS
#define CTGCN_GEMM_NT_LOADS 1        // X planes are streamed once: non-temporal direct-to-LDS loads keep them from evicting W from the XCD's L2
#endif

// gemm_h2_panel_kernel<NT>: Y[M, N] = X W^T for N <= 128 NT.  Persistent blocks of 8 waves (two per SIMD), one block per CU.
//   * A block works on PANELS of 128 rows of X over the full width: X's planes are read from HBM exactly once (round 4's 128 x 128 tiles
//     re-read them N / 128 times through L2, in 64-byte pieces).  k advances in STAGES of 64: per row and plane one whole 128-byte line,
//     fetched by global_load_lds_dwordx4 into a three-slot LDS ring (32 KB per stage; XOR swizzle applied on the global side so that the
//     fragment reads are conflict-free), no registers, no ds_write.
//   * Wave w owns output columns [16 NT w, 16 NT (w + 1)): its W fragments come from the packed layout (above) with one coalesced 1 KB load
//     per (column tile, k slab, plane) STRAIGHT INTO REGISTERS — W is never in LDS, nobody shares it, no barrier guards it.  Every wave reads
//     all 128 X rows of the stage from LDS (8 x 32 KB of ds_read_b128 per stage: 22 % of the LDS read rate at NT = 3).
//   * Products: v_mfma_f32_16x16x32_f16 with W as the first operand, so a lane ends up with FOUR CONSECUTIVE COLUMNS of one output row
//     (16-byte stores); per (row tile, column tile, k slab) three MFMAs into one accumulator, small terms first (x1 w2, x2 w1, x1 w1).
//   * The X requests are INLINE-ASM instructions: behind the compiler's own LDS-DMA intrinsic every ds_read of the ring waits for vmcnt(0) —
//     the compiler cannot tell which slot a request fills — i.e. for requests issued a moment earlier (seen in the ISA).  Their completion is
//     this kernel's business: `s_waitcnt vmcnt(0)` + s_barrier at the top of a stage.  The W fragments are plain loads the compiler tracks.
//   * Requests TRAIL the MFMA groups one or two at a time (a per-stage timeline, tools/gemm_timeline.py: ten requests per wave in one burst
//     behind the barrier kept the CU's address unit busy for ~2 000 cycles during which no wave reached its MFMAs):
//         slab 0:  W fragments of slab 1
//         slab 1:  W fragments of the next stage's slab 0, then the X lines of stage + 2 and the panel's row scales
//     Memory operations retire in order and the compiler does not see the X requests, so both waits of a stage (top, in front of slab 1) are
//     vmcnt(0): the X lines are in LDS 1.5 stages before their use, the W fragments were requested one slab ahead.
//   * A finished panel's rows leave group by group inside the first slab of the NEXT panel (each group right in front of the first products
//     into the same accumulators): the next panel's operands are already in flight.  The stores (16 rows x 64 bytes per instruction) still
//     cost ~1.4 stage times per panel: the remaining known inefficiency (14 % at k = 512).
//   * Registers (NT = 3): 96 accumulators + 48 W fragments (two slabs) + 32 X fragments (two row-tile groups) = 176 + addresses; 251 allocated.
//   * Measured (tools/gemm_bench.py, GEMM alone): 435 180 x 500 x 384 0.55 ms (round 4's 128 x 128 tiles: 0.77), 60 730 x 1 737 x 500 0.27 (0.35),
//     60 730 x 500 x 500 0.10 (0.15); SQ counters: matrix pipe busy 44 % of the SIMD cycles at the 1.85 GHz the chip holds here, waves parked 40 %.
template <int NT, bool GROUPED = false, bool CHAIN = false>
__global__ __launch_bounds__(512, 2) void gemm_h2_panel_kernel(const PanelArgs a)
{
    constexpr int PLANE = PBM * PSK;                      // halfs of one plane of a stage (16 KB)
    __shared__ __attribute__((aligned(1024))) _Float16 As[3][2 * PLANE];      // [slot][plane][row][64 k], 16-byte segment g of row r at g ^ (r & 7)
    __shared__ __attribute__((aligned(16))) float s_sb[2][NT * 128];          // column scales / bias of the group of the current panel and of the one before
    __shared__ __attribute__((aligned(16))) float s_bias[2][NT * 128];
    // CHAIN: a row's scale needs the largest |y| of the row over ALL columns — eight waves' partial maxima, exchanged here ([panel parity][row][wave])
    __shared__ __attribute__((aligned(16))) float s_rmax[CHAIN ? 2 : 1][CHAIN ? PBM : 1][8];
    __shared__ __attribute__((aligned(256))) float s_sa[4][PBM];               // row scales of panel count & 3 (four: one k stage per panel leaves no barrier between
                                                                              // a late wave's epilogue reads and an early wave's request for the panel after next)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;        // wave: in an SGPR
    const int nks = a.Kp / PSK, KS = a.Kp / 32;
    // panels of this block: blockIdx, blockIdx + grid, ... — or, in a grouped launch, a contiguous range (the group changes once or twice
    // per block instead of with every panel)
    constexpr bool grouped = GROUPED;
    const int64_t per_block = (a.mtiles + gridDim.x - 1) / gridDim.x;
    const int64_t first = grouped ? blockIdx.x * per_block : (int64_t)blockIdx.x;
    const int64_t stride = grouped ? 1 : (int64_t)gridDim.x;
    const int64_t pend = grouped ? min(first + per_block, a.mtiles) : a.mtiles;
    if (first >= pend) return;
    const int64_t npan = (pend - first + stride - 1) / stride;
    const int64_t total = npan * nks;

    auto load_scales = [&](int slot_, const float *sb_, const float *bias_) __attribute__((always_inline)) {
        for (int i = tid; i < NT * 128; i += 512) {
            s_sb[slot_][i] = sb_[i];
            s_bias[slot_][i] = (bias_ && i < a.N) ? bias_[i] : 0.f;
        }
    };
    int grp_c = grouped ? a.panel_group[first] : 0;          // group of the panel being multiplied
    int sslot = 0;                                        // its slot of s_sb / s_bias (the panel before: sslot_prev)
    int sslot_prev = 0;
    if (grouped) load_scales(0, a.groups[grp_c].sb, a.groups[grp_c].bias);
    else load_scales(0, a.sb, a.bias);

    // X staging: a wave instruction fills 8 rows x 128 B of one plane linearly (lane l -> row l >> 3, stored segment l & 7), so the lane fetches
    // the segment that belongs there: (l & 7) ^ (row & 7).  32 instructions per stage, four per wave: waves 0-3 plane 1, waves 4-7 plane 2.
    // M0 = LDS address of the 1 KB the instruction fills.
    const _Float16 *aplane = (wave < 4) ? a.a1 : a.a2;
    const int rg0 = (wave & 3) * 32;                      // first of this wave's 32 rows
    const int lrow = lane >> 3;
    const int gseg = ((lane & 7) ^ lrow) * 8;
    const uint32_t ldst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) _Float16 *)&As[0][(wave >> 2) * PLANE + rg0 * PSK];
    auto dma1 = [&](int64_t pn, int ksn, int slot, int i) __attribute__((always_inline)) {
        if ((CTGCN_GEMM_ABLATE & 2) && (pn != first || ksn > 1)) { asm volatile("s_nop 0"); return; }
        const int64_t row = min(pn * PBM + rg0 + i * 8 + lrow, a.M - 1);
        const _Float16 *src = aplane + row * a.Kp + ksn * PSK + gseg;
        const uint32_t dst = ldst + (uint32_t)(slot * 2 * PLANE + i * 8 * PSK) * 2u;
#if CTGCN_GEMM_NT_LOADS
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src), "s"(dst) : "memory");
#else
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory");
#endif
    };
    // the panel's row scales, 4 bytes per lane, straight into s_sa[par]: waves w and w + 2, w + 4, w + 6 fetch the same 64 values (every wave
    // issues the same number of requests: the waits below count them).  Re-requested in every stage of the panel (256 bytes).
    const uint32_t sdst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)&s_sa[0][(wave & 1) * 64];
    auto dma_sa = [&](int64_t pn, int par) __attribute__((always_inline)) {
        const float *src = a.sa + min(pn * PBM + (wave & 1) * 64 + lane, a.M - 1);
        const uint32_t dst = sdst + (uint32_t)par * PBM * 4u;
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(src), "s"(dst) : "memory");
    };

    // W fragments: column tile (wave NT + j), slab s, plane p -> 512 halfs at ((j KS + s) 2 + p) 512 from the wave's base.  Scalar base +
    // one 32-bit lane offset: the address costs no vector registers.  Plain loads the compiler tracks: it waits for them (vmcnt) in front of
    // their first MFMA.  (Inline-asm loads with hand-counted waits were built, measured 2-5 % faster — and gave wrong panels in one run of
    // ten at N <= 128, also with every wait at vmcnt(0): removed, profiles/r05_gemm_panel_kernel.txt.)
    const size_t wave_off = (size_t)wave * NT * KS * 1024;
    const _Float16 *bw_c = (grouped ? a.groups[grp_c].bp : a.bp) + wave_off;         // W fragments of the current panel's group / the next stage's
    const _Float16 *bw_n = bw_c;
    const uint32_t blane = lane * 16;                     // bytes
    h8v fb[2][NT][2];
    auto loadB1 = [&](const _Float16 *bwave, int buf, int s, int q) __attribute__((always_inline)) {         // request q = 2 j + p of slab s
        if ((CTGCN_GEMM_ABLATE & 1) && s > 1) return;
        const int j = q >> 1, p = q & 1;
        fb[buf][j][p] = *(const h8v *)((const char *)(bwave + ((size_t)(j * KS + s) * 2 + p) * 512) + blane);
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
    // four X fragments more), the next group's fragments requested before this group's 3 RG NT products, side(g) — the memory requests that
    // trail group g — behind them.  An accumulator is written by every RG NT-th MFMA: >= 3 issue slots apart
    constexpr int RG = NT >= 4 ? 1 : 2, NG = 8 / RG;
    auto slab = [&](int slot, int sl, int buf, auto pre, auto side) __attribute__((always_inline)) {
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
            pre(rp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int rr = 0; rr < RG; ++rr)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (!(CTGCN_GEMM_ABLATE & 4))
                            acc[rp * RG + rr][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[buf][j][term == 0 ? 1 : 0], xa[cur][rr][term == 1 ? 1 : 0], acc[rp * RG + rr][j], 0, 0, 0);
                        else asm volatile("" :: "v"(fb[buf][j][term == 0 ? 1 : 0]), "v"(xa[cur][rr][term == 1 ? 1 : 0]));
                    }
            __builtin_amdgcn_sched_barrier(0);
            side(rp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // y = act(acc sa[m] sb[n] + bias[n]): lane l holds columns 4 (l >> 4) .. + 3 of row (l & 15) of every 16 x 16 tile.  A panel that lies
    // inside the matrix (rows) whose wave's columns all exist takes the straight-line path — 8 NT 16-byte stores back to back; with a test
    // around every store the compiler branches around each one and waits for vmcnt(0) in front of it (stores count in vmcnt on gfx9).
    auto epilogue_as = [&](int64_t pn, int par, int ss, int r0, int r1, auto selu, auto whole) __attribute__((always_inline)) {
        const int64_t m0 = pn * PBM;
#pragma unroll
        for (int r = r0; r < r1; ++r) {
            const int64_t m = m0 + r * 16 + arow;
            const float s = s_sa[par][r * 16 + arow];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (wave * NT + j) * 16 + 4 * (lane >> 4);
                const f4v sbv = *(const f4v *)&s_sb[ss][n], bv = *(const f4v *)&s_bias[ss][n];
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
                if (CTGCN_GEMM_ABLATE & 8) { asm volatile("" :: "v"(o)); continue; }
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
    auto epilogue = [&](int64_t pn, int par, int ss, int r0, int r1) __attribute__((always_inline)) {
        const bool whole = cols_whole && (pn + 1) * PBM <= a.M;
        if (a.act == 1) {
            if (whole) epilogue_as(pn, par, ss, r0, r1, std::true_type{}, std::true_type{});
            else epilogue_as(pn, par, ss, r0, r1, std::true_type{}, std::false_type{});
        } else {
            if (whole) epilogue_as(pn, par, ss, r0, r1, std::false_type{}, std::true_type{});
            else epilogue_as(pn, par, ss, r0, r1, std::false_type{}, std::false_type{});
        }
    };

    // CHAIN.  transform: when a panel's last product is in, y = act(acc sa sb + bias) replaces the accumulators IN PLACE and every wave leaves the
    // largest |y| of its columns per row in s_rmax (read behind the next barrier).  leave: scale = h2_scale(max over the eight waves), planes =
    // the two fp16 terms of y / scale — the arithmetic of split_rows_h2_kernel on the fp32 row this launch does not write.
    auto chain_transform = [&](int par, int ss, int rbuf) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float s = s_sa[par][r * 16 + arow];
            float m = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (wave * NT + j) * 16 + 4 * (lane >> 4);
                const f4v sbv = *(const f4v *)&s_sb[ss][n], bv = *(const f4v *)&s_bias[ss][n];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float t = fmaf(acc[r][j][v], s * sbv[v], bv[v]);
                    float o = t;
                    if (a.act == 1) {
                        const float neg = (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f) * expm1f(fminf(t, 0.f));
                        o = t > 0.f ? 1.0507009873554804934193349852946f * t : neg;
                    }
                    acc[r][j][v] = o;
                    m = fmaxf(m, fabsf(o));
                }
            }
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            if (lane < 16) s_rmax[rbuf][r * 16 + arow][wave] = m;
        }
    };
    auto chain_leave = [&](int64_t pn, int rbuf, int r0, int r1) __attribute__((always_inline)) {
        const int64_t m0 = pn * PBM;
#pragma unroll
        for (int r = r0; r < r1; ++r) {
            const int64_t m = m0 + r * 16 + arow;
            const f4v ma = *(const f4v *)&s_rmax[rbuf][r * 16 + arow][0], mb = *(const f4v *)&s_rmax[rbuf][r * 16 + arow][4];
            const float rm = fmaxf(fmaxf(fmaxf(ma[0], ma[1]), fmaxf(ma[2], ma[3])), fmaxf(fmaxf(mb[0], mb[1]), fmaxf(mb[2], mb[3])));
            float sc, inv;
            h2_scale(rm, sc, inv);
            if (wave == 0 && lane < 16 && m < a.M) a.osc[m] = sc;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = (wave * NT + j) * 16 + 4 * (lane >> 4);
                h4v hi, lo;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float xs = acc[r][j][v] * inv;
                    hi[v] = (_Float16)xs;
                    lo[v] = (_Float16)(xs - (float)hi[v]);
                }
                acc[r][j] = f4v{0.f, 0.f, 0.f, 0.f};
                if (m < a.M && n < a.okp) {
                    *(h4v *)(a.o1 + m * a.okp + n) = hi;
                    *(h4v *)(a.o2 + m * a.okp + n) = lo;
                }
            }
        }
    };
    int rbuf = 0;                                         // CHAIN: parity of the panel whose maxima sit in s_rmax (flips per finished panel)

    int64_t pan_c = first, pan_2 = first;                 // the stage being multiplied: (panel, k stage); the stage requested two ahead
    int ks_c = 0, ks_2 = 0, par = 0, slot = 0;             // par: (panels finished so far) & 3
    auto advance = [&](int64_t &pn, int &ks) __attribute__((always_inline)) { if (++ks == nks) { ks = 0; pn += stride; } };
    // prologue: W fragments of stage 0 / slab 0, X lines of stages 0 and 1, the first panel's scales — the request pattern of a stage's slab 1
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q) loadB1(bw_c, 0, 0, q);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma1(pan_2, ks_2, 0, i);
    advance(pan_2, ks_2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (also: the LDS stores of s_sb / s_bias above are the compiler's to order: __syncthreads below)
#pragma unroll
    for (int i = 0; i < 4; ++i) dma1(pan_2, ks_2, 1, i);
    advance(pan_2, ks_2);
    dma_sa(pan_c, 0);
    // requests that trail the row-tile groups of a slab: QB per group from a list of 2 NT (slab 0) / 2 NT + 5 (slab 1) requests, early groups first
    constexpr int Q0 = (2 * NT + NG - 1) / NG, Q1 = (2 * NT + 5 + NG - 1) / NG;
    for (int64_t it = 0; it < total; ++it) {
        int64_t pan_n = pan_c;
        int ks_n = ks_c;
        advance(pan_n, ks_n);
#ifdef CTGCN_GEMM_TIMELINE
        unsigned long long *tl = (a.timeline && it < 48 && (wave == 0 || wave == 7) && lane == 0) ? a.timeline + (((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 48 + it) * 6 : nullptr;
        if (tl) tl[0] = clock64();
#endif
        // grouped launch: the next stage's panel may belong to another snapshot (its W), and this stage may be the first of a panel of another
        // snapshot than the panel before (below: its column scales / bias into the other slot of s_sb — the leaving panel's rows need the old ones)
        if (grouped && ks_n == 0 && pan_n < pend) bw_n = a.groups[a.panel_group[pan_n]].bp + wave_off;
        // everything requested so far has landed: the W fragments of this stage's slab 0, this wave's share of the X lines of stages it .. it + 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CTGCN_GEMM_TIMELINE
        if (tl) tl[1] = clock64();
#endif
        __syncthreads();                                  // the stage is complete for every wave; the slot of stage it - 1 has been read by every wave
        if (grouped && ks_c == 0 && it > 0) {             // behind the barrier: no wave is still inside the stores that read the slot written here
            const int g_ = a.panel_group[pan_c];
            sslot_prev = sslot;
            if (g_ != grp_c) { grp_c = g_; sslot ^= 1; load_scales(sslot, a.groups[g_].sb, a.groups[g_].bias); }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef CTGCN_GEMM_TIMELINE
        if (tl) tl[2] = clock64();
#endif
        auto side0 = [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int q = g * Q0; q < (g + 1) * Q0 && q < 2 * NT; ++q) loadB1(bw_c, 1, ks_c * 2 + 1, q);
        };
        // a panel is finished: its rows leave group by group, each right in front of the first products of the NEXT panel into the same
        // accumulators — the stores of rows 32 .. 127 are issued among the MFMAs of rows 0 .. 95 (one block of 8 NT stores in front of
        // the stage cost 1.5 stage times per panel: the address unit takes a 16-row x 64-byte store in no less than a 1 KB load).
        // ONE copy of slab 0 with a uniform branch per group (two copies — with and without the stores — spilled 68 registers at NT = 3)
        const bool leave = ks_c == 0 && it > 0;
        const int64_t pprev = pan_c - stride;
        const int sprev = (par + 3) & 3;
        slab(slot, 0, 0, [&](int g) __attribute__((always_inline)) {
            if (leave) {
                if constexpr (CHAIN) chain_leave(pprev, rbuf ^ 1, g * RG, g * RG + RG);
                else epilogue(pprev, sprev, sslot_prev, g * RG, g * RG + RG);
            }
        }, side0);
#ifdef CTGCN_GEMM_TIMELINE
        if (tl) tl[3] = clock64();
#endif
        // (here the compiler waits for slab 1's W fragments: vmcnt(0) — it does not see the X requests, all older)
#ifdef CTGCN_GEMM_TIMELINE
        if (tl) tl[4] = clock64();
#endif
        const int nslot = slot == 0 ? 2 : slot - 1;       // stage it + 2 goes into the slot of stage it - 1
        slab(slot, 1, 1, [&](int) __attribute__((always_inline)) {}, [&](int g) __attribute__((always_inline)) {
#pragma unroll
            for (int q = g * Q1; q < (g + 1) * Q1 && q < 2 * NT + 5; ++q) {
                if (q < 2 * NT) loadB1(bw_n, 0, ks_n * 2, q);   // the next stage's first slab (past the end: a valid, unused address)
                else if (q < 2 * NT + 4) dma1(pan_2, ks_2, nslot, q - 2 * NT);       // never conditional (past the end: a valid address, a dead slot)
                else dma_sa(pan_n, ks_n == 0 ? (par + 1) & 3 : par);                        // the scales of the NEXT stage's panel
            }
        });
#ifdef CTGCN_GEMM_TIMELINE
        if (tl) tl[5] = clock64();
#endif
        advance(pan_2, ks_2);
        if constexpr (CHAIN) {
            if (ks_n == 0) { chain_transform(par, sslot, rbuf); rbuf ^= 1; }       // the panel is complete: its rows leave in the next stage
        }
        if (ks_n == 0) par = (par + 1) & 3;
        bw_c = bw_n;
        pan_c = pan_n;
        ks_c = ks_n;
        slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (CHAIN) chain_leave(pan_c - stride, rbuf ^ 1, 0, 8);
    else epilogue(pan_c - stride, (par + 3) & 3, sslot, 0, 8);
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
#ifdef CTGCN_GEMM_TIMELINE
    static const char *tl_file = getenv("CTGCN_GEMM_TIMELINE_FILE");
    static int tl_calls = 0;
    const size_t tl_words = (size_t)blocks * 2 * 48 * 6;
    if (tl_file && ++tl_calls == 3) {            // the third call: clocks and caches are warm
        (void)hipMalloc(&a.timeline, tl_words * 8);
        (void)hipMemsetAsync(a.timeline, 0, tl_words * 8, (hipStream_t)stream);
    }
#endif
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
#ifdef CTGCN_GEMM_TIMELINE
    if (a.timeline) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        unsigned long long *h = (unsigned long long *)malloc(tl_words * 8);
        (void)hipMemcpy(h, a.timeline, tl_words * 8, hipMemcpyDeviceToHost);
        FILE *f = fopen(tl_file, "w");
        for (size_t i = 0; i < tl_words / 6; ++i)
            if (h[i * 6]) fprintf(f, "%zu %zu %zu %llu %llu %llu %llu %llu %llu\n", i / 96, (i / 48) % 2, i % 48, h[i * 6], h[i * 6 + 1], h[i * 6 + 2], h[i * 6 + 3], h[i * 6 + 4], h[i * 6 + 5]);
        fclose(f);
        free(h);
        (void)hipFree(a.timeline);
    }
#endif
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_linear_packed_chain_f32(int64_t rows, int32_t n_out, int32_t k, const void *x_planes, const void *w_packed, const float *bias, int32_t activation,
                                  void *out_planes, size_t out_planes_bytes, void *stream)
{
    if (rows < 0 || n_out < 1 || n_out > PCHUNK || k < 1) return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed_chain: bad sizes (n_out <= 512)");
    if (activation != CTGCN_ACT_NONE && activation != CTGCN_ACT_SELU) return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed_chain: unknown activation");
    if (rows == 0) return CTGCN_OK;
    if (!x_planes || !w_packed || !out_planes || (reinterpret_cast<uintptr_t>(x_planes) & 255u) || (reinterpret_cast<uintptr_t>(w_packed) & 255u) ||
        (reinterpret_cast<uintptr_t>(out_planes) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed_chain: null or misaligned (256 bytes) operand buffers");
    if (out_planes_bytes < ctgcn_split_planes_bytes(rows, n_out)) return ctgcn_set_error_(CTGCN_E_WORKSPACE, "linear_packed_chain: out_planes too small (ctgcn_split_planes_bytes(rows, n_out))");
    const PackGeom g = pack_geom(n_out, k);
    PanelArgs a{};
    a.M = rows; a.N = n_out; a.Kp = g.kp;
    a.a1 = (const _Float16 *)x_planes; a.a2 = a.a1 + (size_t)rows * g.kp; a.sa = (const float *)(a.a2 + (size_t)rows * g.kp);
    a.act = activation;
    a.mtiles = (rows + PBM - 1) / PBM;
    const _Float16 *frags = (const _Float16 *)w_packed;
    a.bp = frags; a.sb = (const float *)(frags + g.frag_halfs); a.bias = bias;
    a.okp = (int32_t)align_up((size_t)n_out, PSK);
    a.o1 = (_Float16 *)out_planes; a.o2 = a.o1 + (size_t)rows * a.okp; a.osc = (float *)(a.o2 + (size_t)rows * a.okp);
    const int64_t blocks = a.mtiles < device_cus() ? a.mtiles : device_cus();
    const dim3 grid((unsigned)blocks), blk(512);
    switch (chunk_nt(n_out)) {
    case 1: hipLaunchKernelGGL((gemm_h2_panel_kernel<1, false, true>), grid, blk, 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL((gemm_h2_panel_kernel<2, false, true>), grid, blk, 0, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL((gemm_h2_panel_kernel<3, false, true>), grid, blk, 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL((gemm_h2_panel_kernel<4, false, true>), grid, blk, 0, (hipStream_t)stream, a); break;
    }
    GEMM_TRY(hipGetLastError());
    return CTGCN_OK;
}

int ctgcn_linear_packed_group_f32(int32_t groups, int64_t total_rows, int32_t n_out, int32_t k, const void *planes1, const void *planes2,
                                  const float *scales, const int32_t *panel_group, const void *const *w_packed, const float *const *bias,
                                  int32_t activation, float *y, int64_t ldy, void *table, size_t table_bytes, void *shadow, void *stream)
{
    if (groups < 1 || groups > 1024 || total_rows < 0 || (total_rows % PBM) || n_out < 1 || n_out > PCHUNK || k < 1 || ldy < n_out)
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed_group: bad sizes (rows in whole panels of 128, n_out <= 512)");
    if (activation != CTGCN_ACT_NONE && activation != CTGCN_ACT_SELU) return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed_group: unknown activation");
    if (total_rows == 0) return CTGCN_OK;
    if (!planes1 || !planes2 || !scales || !panel_group || !w_packed || !y || (reinterpret_cast<uintptr_t>(planes1) & 255u) ||
        (reinterpret_cast<uintptr_t>(planes2) & 255u) || (reinterpret_cast<uintptr_t>(scales) & 255u))
        return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed_group: null or misaligned (256 bytes) operand buffers");
    if (!table || (reinterpret_cast<uintptr_t>(table) & 255u) || table_bytes < (size_t)groups * sizeof(PanelGroup))
        return ctgcn_set_error_(CTGCN_E_WORKSPACE, "linear_packed_group: table must be 256-byte aligned and hold 24 bytes per group");
    const PackGeom g = pack_geom(n_out, k);
    std::vector<PanelGroup> host((size_t)groups);
    for (int i = 0; i < groups; ++i) {
        if (!w_packed[i] || (reinterpret_cast<uintptr_t>(w_packed[i]) & 255u))
            return ctgcn_set_error_(CTGCN_E_INVALID, "linear_packed_group: null or misaligned packed weight");
        const _Float16 *frags = (const _Float16 *)w_packed[i];
        host[i] = PanelGroup{frags, (const float *)(frags + g.frag_halfs), bias ? bias[i] : nullptr};
    }
    GEMM_TRY(ctgcn_table::upload(table, host.data(), host.size() * sizeof(PanelGroup), shadow, (hipStream_t)stream));
    PanelArgs a{};
    a.M = total_rows; a.N = n_out; a.Kp = g.kp;
    a.a1 = (const _Float16 *)planes1; a.a2 = (const _Float16 *)planes2; a.sa = scales;
    a.act = activation; a.ldy = ldy; a.y = y;
    a.vec = (!(ldy & 3) && !(reinterpret_cast<uintptr_t>(y) & 15u)) ? 1 : 0;
    a.mtiles = total_rows / PBM;
    a.panel_group = panel_group; a.groups = (const PanelGroup *)table;
    a.bp = host[0].bp; a.sb = host[0].sb; a.bias = host[0].bias;
    const int64_t blocks = a.mtiles < device_cus() ? a.mtiles : device_cus();
    const dim3 grid((unsigned)blocks), blk(512);
    switch (chunk_nt(n_out)) {
    case 1: hipLaunchKernelGGL((gemm_h2_panel_kernel<1, true>), grid, blk, 0, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL((gemm_h2_panel_kernel<2, true>), grid, blk, 0, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL((gemm_h2_panel_kernel<3, true>), grid, blk, 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL((gemm_h2_panel_kernel<4, true>), grid, blk, 0, (hipStream_t)stream, a); break;
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
