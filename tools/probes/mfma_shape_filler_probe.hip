// Probe (round 3): how many VALU instructions hide behind one MFMA on gfx950, per MFMA shape and waves per SIMD?
// The GRU layer kernel issues v_mfma_f32_16x16x32_f16 (16 cycles of matrix pipe each) and ~100 VALU per 72 MFMAs per wave; round 2
// measured that at two waves per SIMD those VALU add their full issue time.  MI355X_MICROARCH.md reports <= 5 single-issue
// fillers hidden per v_mfma_f32_32x32x16 (32 cycles of matrix pipe) at one wave per SIMD.  This probe measures both shapes at
// equal matrix work (2 x 16x16x32 = 1 x 32x32x16) with K fillers per 32 cycles of matrix pipe, 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sf tools/probes/mfma_shape_filler_probe.hip && /tmp/sf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// SHAPE 0: 16x16x32 (two per slot), 1: 32x32x16 (one per slot).  K fillers per slot (= 32 matrix-pipe cycles).  TRANS: v_exp instead of v_fma.
template <int SHAPE, int K, int TRANS, int NT>
__global__ __launch_bounds__(NT) void k(float *out, int iters, float scale)
{
    h8v a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) {
        a[i][j] = (_Float16)(scale * (float)(((threadIdx.x * 7 + i * 3 + j) % 13) - 6));
        b[i][j] = (_Float16)(scale * (float)(((threadIdx.x * 5 + i + j * 3) % 11) - 5));
    }
    f4v acc4[6];
    f16v acc16[3];
    for (int i = 0; i < 6; ++i) acc4[i] = f4v{0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    const float c1 = 0.999f, c2 = 1e-4f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 12; ++m) {       // 12 slots of 32 matrix-pipe cycles
            if (SHAPE == 0) {
                acc4[(2 * m) % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m & 3], b[(m >> 2) & 3], acc4[(2 * m) % 6], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < K / 2; ++q) {
                    float &x = v[(m * K + q) & 7];
                    if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
                }
                acc4[(2 * m + 1) % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(m + 1) & 3], b[(m >> 2) & 3], acc4[(2 * m + 1) % 6], 0, 0, 0);
#pragma unroll
                for (int q = K / 2; q < K; ++q) {
                    float &x = v[(m * K + q) & 7];
                    if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
                }
            } else {
                acc16[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 3], b[(m >> 2) & 3], acc16[m % 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    float &x = v[(m * K + q) & 7];
                    if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
                }
            }
        }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) s += acc4[i][0];
    for (int i = 0; i < 3; ++i) s += acc16[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int SHAPE, int K, int TRANS, int NT>
static void run(float *out, int iters, const char *what)
{
    k<SHAPE, K, TRANS, NT><<<256, NT>>>(out, 10, 0.37f);
    hipDeviceSynchronize();
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(s); k<SHAPE, K, TRANS, NT><<<256, NT>>>(out, iters, 0.37f); hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        best = ms < best ? ms : best;
    }
    const double slots = (double)iters * 12;
    // per SIMD: NT / 256 waves share it; ns per slot per SIMD = time / slots / waves
    printf("%-14s %d wave(s)/SIMD  %d %s per 32 pipe-cycles : %8.3f ms = %6.2f ns per slot per wave = %6.2f ns per slot per SIMD\n", SHAPE ? "32x32x16_f16" : "2x 16x16x32_f16",
           NT / 256, K, TRANS ? "v_exp" : "v_fma", best, best * 1e6 / slots, best * 1e6 / slots / (NT / 256));
}

int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
#define ROW(S, T, NT) run<S, 0, T, NT>(out, iters, ""); run<S, 2, T, NT>(out, iters, ""); run<S, 4, T, NT>(out, iters, ""); run<S, 6, T, NT>(out, iters, ""); run<S, 8, T, NT>(out, iters, ""); run<S, 12, T, NT>(out, iters, "");
    ROW(0, 0, 256) ROW(1, 0, 256) ROW(0, 0, 512) ROW(1, 0, 512)
    ROW(0, 1, 256) ROW(1, 1, 256) ROW(0, 1, 512) ROW(1, 1, 512)
    return 0;
}
