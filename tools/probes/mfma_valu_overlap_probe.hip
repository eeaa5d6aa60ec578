// Probe: do VALU instructions run concurrently with the matrix pipe on gfx950?
// One block of 512 threads per CU (2 waves/SIMD, as in the GRU kernels).  Each wave issues a stream of
// v_mfma_f32_16x16x32_bf16 (3 independent accumulators) with K independent VALU ops (v_fma_f32 or v_exp_f32) slotted in
// after every MFMA.  If the pipes overlap, time stays at the MFMA-only time until K*4 cycles exceed the 16-cycle MFMA;
// if they serialise, time grows by 4 cycles (fma) per VALU op from K = 1.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/ov tools/probes/mfma_valu_overlap_probe.hip && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int K, int TRANS, int MFMA>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters)
{
    bf8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(float)(threadIdx.x + i + j); b[i][j] = (__bf16)(float)(threadIdx.x * 3 + i - j); }
    f4v acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    const float c1 = 0.999f, c2 = 1e-4f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (MFMA) acc[m % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 3], b[(m >> 2) & 3], acc[m % 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                float &x = v[(m * K + q) & 7];
                if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
            }
        }
    float s = acc[0][0] + acc[1][1] + acc[2][2];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int TRANS, int MFMA>
static float run(float *out, int iters)
{
    k<K, TRANS, MFMA><<<256, 512>>>(out, 10);
    hipDeviceSynchronize();
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipEventRecord(s); k<K, TRANS, MFMA><<<256, 512>>>(out, iters); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms;
}

int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 4000;
    const double n = (double)iters * 24;      // MFMAs per wave
    auto rep = [&](const char *name, float ms) { printf("%-34s %8.3f ms  = %6.1f ns per MFMA slot per wave\n", name, ms, ms * 1e6 / n); };
    rep("MFMA only", run<0, 0, 1>(out, iters));
    rep("MFMA + 1 fma", run<1, 0, 1>(out, iters));
    rep("MFMA + 2 fma", run<2, 0, 1>(out, iters));
    rep("MFMA + 3 fma", run<3, 0, 1>(out, iters));
    rep("MFMA + 4 fma", run<4, 0, 1>(out, iters));
    rep("MFMA + 6 fma", run<6, 0, 1>(out, iters));
    rep("MFMA + 8 fma", run<8, 0, 1>(out, iters));
    rep("no MFMA, 3 fma", run<3, 0, 0>(out, iters));
    rep("no MFMA, 8 fma", run<8, 0, 0>(out, iters));
    rep("MFMA + 1 exp", run<1, 1, 1>(out, iters));
    rep("MFMA + 2 exp", run<2, 1, 1>(out, iters));
    rep("MFMA + 4 exp", run<4, 1, 1>(out, iters));
    rep("no MFMA, 2 exp", run<2, 1, 0>(out, iters));
    rep("no MFMA, 4 exp", run<4, 1, 0>(out, iters));
    return 0;
}
