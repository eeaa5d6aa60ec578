// Probe: two waves on one SIMD, one issuing only MFMAs, the other only VALU work (the gate math of a GRU step: v_fma_f32 /
// v_exp_f32 / v_rcp_f32 mix) — do they run concurrently?  (mfma_valu_overlap_probe.hip interleaves both kinds inside EACH
// wave and measured the sum of the two.)  Blocks of 512 threads, one per CU: waves 0-3 and 4-7 share SIMDs 0-3.
//   mode 0: all 8 waves MFMA only            mode 1: all 8 waves VALU only
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU   (same per-wave work as in modes 0 / 1)
//   mode 3: every wave MFMA then VALU (phases, all waves in step — what gru_layer8_h2_kernel does today)
// If the pipes overlap across waves, mode 2 takes max(mode 0, mode 1) / 2-ish; if not, their mean.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/mfma_valu_two_waves_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

template <int NM>
__device__ __forceinline__ void mfma_burst(f4v (&acc)[6], const h8v (&a)[2], const h8v (&b)[2])
{
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m % 6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m & 1], b[(m >> 1) & 1], acc[m % 6], 0, 0, 0);
}

template <int NV>
__device__ __forceinline__ void valu_burst(float (&v)[8], float c1, float c2)
{
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        float &x = v[q & 7];
        if (q % 8 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else if (q % 8 == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    }
}

__global__ __launch_bounds__(512, 2) void k(float *out, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    h8v a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(0.01f * ((threadIdx.x * 7 + i + j) % 13 - 6)); b[i][j] = (_Float16)(0.02f * ((threadIdx.x * 3 + i - j) % 11 - 5)); }
    f4v acc[6] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    const float c1 = 0.999f, c2 = 1e-4f;
    constexpr int NM = 72, NV = 160;          // per unit and wave: the MFMAs and roughly the VALU instructions of gru_layer8_h2_kernel
    for (int it = 0; it < iters; ++it) {
        const bool do_m = mode == 0 || mode == 3 || (mode == 2 && wave < 4) || (mode == 4 && ((it + (wave >> 2)) & 1) == 0);
        const bool do_v = mode == 1 || mode == 3 || (mode == 2 && wave >= 4) || (mode == 4 && ((it + (wave >> 2)) & 1) == 1);
        if (do_m) mfma_burst<NM>(acc, a, b);
        if (do_v) valu_burst<NV>(v, c1, c2);
        if (mode == 3 || mode == 4) __syncthreads();
    }
    float r = 0.f;
    for (int i = 0; i < 6; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) r += v[i];
    if (r == 123.456f) out[0] = r;
}

int main()
{
    float *out;
    hipMalloc(&out, 4);
    const int iters = 2000, blocks = 256;
    const char *names[] = {"all waves: 72 MFMAs per unit", "all waves: 160 VALU per unit", "waves 0-3 MFMA, 4-7 VALU (no barrier)",
                           "every wave MFMA then VALU, barrier per unit (today)", "wave groups alternate MFMA / VALU units, barrier per unit"};
    for (int mode = 0; mode < 5; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, 50, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d  %-62s %8.3f ms  = %7.1f ns per unit\n", mode, names[mode], ms, ms * 1e6 / iters);
    }
    return 0;
}
