// Probe: fp32 GEMM tile via 3-way bf16 split (6 products) on v_mfma_f32_16x16x32_bf16; checks layout + accuracy + rate.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/mfma_bf16x3_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16 &a, __bf16 &b, __bf16 &c)
{
    a = (__bf16)x; float r = x - (float)a;
    b = (__bf16)r; float r2 = r - (float)b;
    c = (__bf16)r2;
}

// C[16x16] = A[16x128] * B[128x16], one wave.  A row-major [16][128], Bt row-major [16 cols][128 k].
__global__ void tile(const float *A, const float *Bt, float *C, int mode)
{
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f4v acc = {0, 0, 0, 0};
    for (int c = 0; c < 4; ++c) {
        bf8 a[3], b[3];
        for (int j = 0; j < 8; ++j) {
            __bf16 x0, x1, x2, y0, y1, y2;
            split3(A[r * 128 + c * 32 + 8 * g + j], x0, x1, x2);
            split3(Bt[r * 128 + c * 32 + 8 * g + j], y0, y1, y2);
            a[0][j] = x0; a[1][j] = x1; a[2][j] = x2; b[0][j] = y0; b[1][j] = y1; b[2][j] = y2;
        }
        const int pi[6] = {2, 1, 0, 0, 1, 0}, pj[6] = {0, 1, 2, 1, 0, 0};      // small terms first
        const int np = mode == 0 ? 6 : (mode == 1 ? 3 : 1);
        for (int p = 6 - np; p < 6; ++p) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pi[p]], b[pj[p]], acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}

__global__ __launch_bounds__(512, 2) void rate(float *out, int iters)
{
    bf8 a[4], b[12];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)(float)(threadIdx.x + i + j);
    for (int i = 0; i < 12; ++i) for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)(float)(threadIdx.x * 3 + i - j);
    f4v acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 24; ++k)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k & 3], b[(k + 4 * g) % 12], acc[g], 0, 0, 0);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2];
}

int main()
{
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A(16 * 128), Bt(16 * 128), C(256);
    for (auto &v : A) v = fabsf(nd(rng)) * 3.f;
    for (auto &v : Bt) v = nd(rng) * 0.09f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        tile<<<1, 64>>>(dA, dB, dC, mode);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double emax = 0, e32 = 0, rmax = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ref = 0; float f = 0;
            for (int k = 0; k < 128; ++k) { ref += (double)A[i * 128 + k] * Bt[j * 128 + k]; f = fmaf(A[i * 128 + k], Bt[j * 128 + k], f); }
            emax = fmax(emax, fabs(C[i * 16 + j] - ref)); e32 = fmax(e32, fabs(f - ref)); rmax = fmax(rmax, fabs(ref));
        }
        printf("mode %d (%s): max|err| vs fp64 = %.3e   (fp32 fmaf chain: %.3e, max|ref| = %.2f)\n", mode,
               mode == 0 ? "6 products" : mode == 1 ? "3 products" : "bf16 only", emax, e32, rmax);
    }
    float *out; hipMalloc(&out, 256 * 512 * 4);
    rate<<<256, 512>>>(out, 10); hipDeviceSynchronize();
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 4000;
    hipEventRecord(s); rate<<<256, 512>>>(out, iters); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double mf = 256.0 * 8 * iters * 72;
    printf("bf16 16x16x32 stream (3 acc, 2 waves/SIMD): %.3f ms, %.1f bf16 TF/s, %.1f cycles/MFMA/SIMD @2.4GHz, fp32-equivalent (6 products) %.1f TF/s\n",
           ms, mf * 16384 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024), mf * 16384 / 6 / ms / 1e9);
    return 0;
}
