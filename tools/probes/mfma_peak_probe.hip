// Probe: what the 16-bit matrix pipe of this MI355X sustains, to price the MFMA-bound kernels against a MEASURED ceiling.
//   * v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16, back to back, 4 independent accumulators per wave
//   * 1, 2 and 4 waves per SIMD (blocks of 256 / 512 / 1024 threads, one block per CU, 256 CUs x 8 blocks deep)
//   * operands of random-ish magnitude (power draw depends on the data; all-zero operands run faster than real ones)
//   * shader clock during the run: clock64() ticks / wall_clock64() ticks (100 MHz)
// hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] -o /tmp/p tools/probes/mfma_peak_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ void burn(int iters, float seed, float *out, unsigned long long *clk)
{
    h8v a[2], b[2];
    for (int p = 0; p < 2; ++p)
        for (int j = 0; j < 8; ++j) {
            a[p][j] = (_Float16)(seed * (float)((threadIdx.x * 7 + j * 3 + p) % 13 - 6));
            b[p][j] = (_Float16)(seed * (float)((threadIdx.x * 5 + j * 11 + p) % 17 - 8));
        }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float r = 0.f;
    if (KIND == 0) {
        f16v acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q & 1], b[q >> 1], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 4; ++q)
            for (int v = 0; v < 16; ++v) r += acc[q][v];
    } else {
        f4v acc[8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q & 1], b[(q >> 1) & 1], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q)
            for (int v = 0; v < 4; ++v) r += acc[q][v];
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    if (r == 123.456f) out[0] = r;
}

template <int KIND>
void run(const char *name, int threads, float seed)
{
    float *out;
    unsigned long long *clk, h[2];
    hipMalloc(&out, 4);
    hipMalloc(&clk, 16);
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(burn<KIND>, dim3(blocks), dim3(threads), 0, 0, 100, seed, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(burn<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, seed, out, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double per = KIND == 0 ? 24.0 * 32768 : 48.0 * 16384;      // flops per wave per iteration
    const double fl = per * iters * (threads / 64) * (double)blocks;
    printf("%-26s %4d threads/block (%d waves/SIMD) operands x%-6g: %7.1f TFLOP/s  (%.2f ms), shader clock %.0f MHz\n", name, threads,
           threads / 256, seed, fl / ms / 1e9, ms, 100.0 * (double)h[0] / (double)h[1]);
    hipFree(out);
    hipFree(clk);
}

int main()
{
    for (float seed : {0.f, 1.f, 0.37f}) {
        for (int threads : {256, 512, 1024}) {
            run<0>("v_mfma_f32_32x32x16_f16", threads, seed);
            run<1>("v_mfma_f32_16x16x32_f16", threads, seed);
        }
    }
    return 0;
}
