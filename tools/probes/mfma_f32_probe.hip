// Throughput probe: v_mfma_f32_16x16x4_f32 streams shaped like gru_seq_kernel's inner loop.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_f32_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));

template <int NACC, int KK>
__global__ __launch_bounds__(512, 2) void probe(float *out, const float *in, int iters)
{
    float W[NACC][KK], av[KK];
    for (int g = 0; g < NACC; ++g)
        for (int k = 0; k < KK; ++k) W[g][k] = in[(g * KK + k) * 64 + (threadIdx.x & 63)];
    for (int k = 0; k < KK; ++k) av[k] = in[k * 64 + (threadIdx.x & 63)] * 0.5f;
    f4v acc[NACC];
    for (int g = 0; g < NACC; ++g) acc[g] = f4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < KK; ++k)
#pragma unroll
            for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], W[g][k], acc[g], 0, 0, 0);
    }
    float s = 0.f;
    for (int g = 0; g < NACC; ++g) s += acc[g].x + acc[g].y + acc[g].z + acc[g].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int KK>
void run(int threads, const char *name)
{
    float *in, *out;
    hipMalloc(&in, NACC * KK * 64 * 4 + 4096);
    hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(in, 0, NACC * KK * 64 * 4 + 4096);
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    probe<NACC, KK><<<256, threads>>>(out, in, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<NACC, KK><<<256, threads>>>(out, in, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double flops = 256.0 * (threads / 64) * iters * NACC * KK * 2048.0;
    printf("%-28s threads/block=%4d  %.3f ms  %.1f TF/s\n", name, threads, ms, flops / ms / 1e9);
    hipFree(in); hipFree(out);
}

int main()
{
    run<3, 32>(512, "3 acc x 32 k (gru shape)");
    run<3, 32>(256, "3 acc x 32 k");
    run<6, 16>(512, "6 acc x 16 k");
    run<6, 16>(256, "6 acc x 16 k");
    run<12, 8>(512, "12 acc x 8 k");
    run<12, 8>(256, "12 acc x 8 k");
    run<4, 8>(256, "4 acc x 8 k");
    run<4, 8>(1024, "4 acc x 8 k");
    return 0;
}
