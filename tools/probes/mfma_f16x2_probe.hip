// Probe: fp32 GEMM tile via a 2-way fp16 split with scaled residuals (3 products, 2 accumulators) on
// v_mfma_f32_16x16x32_f16, against the 3-way bf16 split (6 products) and an fp32 fmaf chain; error vs fp64.
//   x = s * (x1 + x2 * 2^-11),  x1 = fp16(x / s),  x2 = fp16((x / s - x1) * 2^11),  s = power of two >= row max / 2^14
//   x.y ~ sx sy [ x1 y1 + 2^-11 (x1 y2 + x2 y1) ]        (dropped: x2 y2 2^-22)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/probes/mfma_f16x2_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16 &a, __bf16 &b, __bf16 &c)
{
    a = (__bf16)x; float r = x - (float)a;
    b = (__bf16)r; float r2 = r - (float)b;
    c = (__bf16)r2;
}
__device__ __forceinline__ void split2h(float x, _Float16 &a, _Float16 &b)
{
    a = (_Float16)x;
    b = (_Float16)((x - (float)a) * 2048.f);
}

// C[16x16] = A[16xK] * B[Kx16], one wave, K = 128.  A row-major [16][128], Bt row-major [16 cols][128 k].  sa/sb: per-row scales.
__global__ void tile(const float *A, const float *Bt, const float *sa, const float *sb, float *C, int mode)
{
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f4v acc = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int c = 0; c < 4; ++c) {
        if (mode == 0) {
            bf8 a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                __bf16 x0, x1, x2, y0, y1, y2;
                split3(A[r * 128 + c * 32 + 8 * g + j], x0, x1, x2);
                split3(Bt[r * 128 + c * 32 + 8 * g + j], y0, y1, y2);
                a[0][j] = x0; a[1][j] = x1; a[2][j] = x2; b[0][j] = y0; b[1][j] = y1; b[2][j] = y2;
            }
            const int pi[6] = {2, 1, 0, 0, 1, 0}, pj[6] = {0, 1, 2, 1, 0, 0};
            for (int p = 0; p < 6; ++p) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pi[p]], b[pj[p]], acc, 0, 0, 0);
        } else {
            h8 a[2], b[2];
            const float ia = 1.f / sa[r], ib = 1.f / sb[r];
            for (int j = 0; j < 8; ++j) {
                _Float16 x1, x2, y1, y2;
                split2h(A[r * 128 + c * 32 + 8 * g + j] * ia, x1, x2);
                split2h(Bt[r * 128 + c * 32 + 8 * g + j] * ib, y1, y2);
                a[0][j] = x1; a[1][j] = x2; b[0][j] = y1; b[1][j] = y2;
            }
            if (mode == 2) {       // residuals NOT rescaled: one accumulator
                for (int j = 0; j < 8; ++j) {
                    const float xa = A[r * 128 + c * 32 + 8 * g + j] * ia, xb = Bt[r * 128 + c * 32 + 8 * g + j] * ib;
                    a[1][j] = (_Float16)(xa - (float)a[0][j]); b[1][j] = (_Float16)(xb - (float)b[0][j]);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
            } else {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
            }
        }
    }
    // D layout: lane (r = n index, g) holds D[m = 4g + i][n = r]
    for (int i = 0; i < 4; ++i) {
        float v = acc[i];
        if (mode == 1) v = (acc[i] + acc1[i] * (1.f / 2048.f)) * sa[4 * g + i] * sb[r];
        if (mode == 2) v = acc[i] * sa[4 * g + i] * sb[r];
        C[(4 * g + i) * 16 + r] = v;
    }
}

static float pow2_scale(const float *row, int n)
{
    float m = 0;
    for (int i = 0; i < n; ++i) m = fmaxf(m, fabsf(row[i]));
    if (m == 0) return 1.f;
    int e; frexpf(m, &e);              // m = f * 2^e, f in [0.5, 1)
    return ldexpf(1.f, e - 15);        // m / s in [2^14, 2^15)
}

int main()
{
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    float *dA, *dB, *dC, *dsa, *dsb;
    hipMalloc(&dA, 16 * 128 * 4); hipMalloc(&dB, 16 * 128 * 4); hipMalloc(&dC, 1024); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64);
    const char *names[4] = {"relu(N)*3 x N*0.09 (projection-like)", "tanh(N) x N*0.09 (recurrence-like)", "wide dynamic range (1e-4..1e2, signed)", "tiny values (|x| ~ 1e-6)"};
    for (int data = 0; data < 4; ++data) {
        double e_bf = 0, e_h = 0, e_h1 = 0, e_32 = 0, rmax = 0;
        for (int trial = 0; trial < 200; ++trial) {
            std::vector<float> A(16 * 128), Bt(16 * 128), C(256), sa(16), sb(16);
            for (auto &v : A) {
                float z = nd(rng);
                v = data == 0 ? fabsf(z) * 3.f * (rng() % 3 != 0) : data == 1 ? tanhf(z) : data == 2 ? powf(10.f, -4.f + 6.f * (rng() % 1000) / 1000.f) * (z > 0 ? 1 : -1) : z * 1e-6f;
            }
            for (auto &v : Bt) v = nd(rng) * 0.09f;
            for (int i = 0; i < 16; ++i) { sa[i] = pow2_scale(&A[i * 128], 128); sb[i] = pow2_scale(&Bt[i * 128], 128); }
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice);
            hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
            for (int mode = 0; mode < 3; ++mode) {
                tile<<<1, 64>>>(dA, dB, dsa, dsb, dC, mode);
                hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
                for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                    double ref = 0, mag = 0; float f = 0;
                    for (int k = 0; k < 128; ++k) { ref += (double)A[i * 128 + k] * Bt[j * 128 + k]; mag += fabs((double)A[i * 128 + k] * Bt[j * 128 + k]); f = fmaf(A[i * 128 + k], Bt[j * 128 + k], f); }
                    // error relative to sum |a_k b_k| (the quantity every dot-product error bound is stated in)
                    const double e = fabs(C[i * 16 + j] - ref) / mag;
                    if (mode == 0) { e_bf = fmax(e_bf, e); e_32 = fmax(e_32, fabs(f - ref) / mag); } else if (mode == 1) e_h = fmax(e_h, e); else e_h1 = fmax(e_h1, e);
                    rmax = fmax(rmax, fabs(ref));
                }
            }
        }
        printf("%-42s max err / sum|ab|:  bf16x3 (6 MFMA) %.2e   fp16x2 (3 MFMA, scaled residual) %.2e   fp16x2 (3 MFMA, one acc) %.2e   fp32 fmaf chain %.2e\n", names[data], e_bf, e_h, e_h1, e_32);
    }
    return 0;
}
