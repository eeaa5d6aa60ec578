// ds_read_b64_tr_b16 semantics probe (gfx950): LDS holds u16 element e at element index e; every lane supplies a byte address; the
// four 16-bit values a lane receives tell which (supplying lane, sub-element) the hardware routed to it.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/ds_read_tr_probe.hip -o tools/probes/bin/ds_read_tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint32_t *addr, uint16_t *out)
{
    __shared__ uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = addr[threadIdx.x] + (uint32_t)(uintptr_t)lds;
    uint32_t lo, hi;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    lo = v[0]; hi = v[1];
    out[threadIdx.x * 4 + 0] = lo & 0xffff; out[threadIdx.x * 4 + 1] = lo >> 16;
    out[threadIdx.x * 4 + 2] = hi & 0xffff; out[threadIdx.x * 4 + 3] = hi >> 16;
}

int main()
{
    uint32_t *d_addr; uint16_t *d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 64 * 4 * 2);
    const char *names[] = {"uniform 0", "lane*8 (lane-linear)", "row-major [k][n] pitch 64 B: k = (l&15)>>2 + 4*(l>>4), n0 = 4*(l&3)",
                           "row-major pitch 64 B: k = l&15 (one row per lane), group g -> cols 4g", "pitch 256 B: k=(l&15)>>2 + 4*(l>>4), n0=4*(l&3)"};
    for (int p = 0; p < 5; ++p) {
        std::vector<uint32_t> a(64);
        for (int l = 0; l < 64; ++l) {
            switch (p) {
            case 0: a[l] = 0; break;
            case 1: a[l] = l * 8; break;
            case 2: a[l] = ((((l & 15) >> 2) + 4 * (l >> 4)) * 32 + 4 * (l & 3)) * 2; break;
            case 3: a[l] = ((l & 15) * 32 + 4 * (l >> 4)) * 2; break;
            case 4: a[l] = ((((l & 15) >> 2) + 4 * (l >> 4)) * 128 + 4 * (l & 3)) * 2; break;
            }
        }
        hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        std::vector<uint16_t> o(256);
        hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d: %s\n", p, names[p]);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr(elem) %4u -> %4u %4u %4u %4u", l, a[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
            if (l % 2 == 1) printf("\n");
        }
    }
    return 0;
}
