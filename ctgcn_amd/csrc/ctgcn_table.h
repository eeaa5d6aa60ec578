// Descriptor tables of the grouped launches (ctgcn_*_group_f32): how the host array reaches the device.
//
// Rounds 4-5 copied the table with hipMemcpyAsync from a function-local pageable vector on every call: the runtime stages such a copy
// before it returns (a host stall of tens of microseconds per call, several calls per layer in a 1.3 ms window), and a host-to-device
// copy cannot be recorded into a hipGraph.  Now
//   * the bytes travel as KERNEL ARGUMENTS of table_write_kernel (<= 3.5 KB per launch): asynchronous, stream-ordered, capturable, no
//     staging buffer and no pinned memory;
//   * a caller that keeps the table alive between calls passes a host SHADOW of it (same size, zero-filled once): when the descriptors
//     of this call equal the shadow, the device copy is current and nothing is written at all — the steady state of an inference loop
//     over one (model, window), whose buffers the caching allocator hands out at the same addresses forward after forward.
// The comparison is on the descriptor bytes themselves (every pointer, size and flag the kernels will read), so a stale table cannot be
// launched; what the caller guarantees is only that nobody else writes to `table` while it holds the shadow.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>

extern "C" void ctgcn_table_count_(int written);       // ctgcn_hip.hip: the diagnostic counters behind ctgcn_table_uploads()

namespace ctgcn_table {

constexpr int CHUNK_BYTES = 3584;                  // kernel arguments live in a 4 KB segment; leave room for the two other arguments
struct Chunk { uint32_t w[CHUNK_BYTES / 4]; };

static __global__ __launch_bounds__(256) void table_write_kernel(Chunk c, uint32_t *__restrict__ dst, int32_t words)
{
    for (int i = threadIdx.x; i < words; i += 256) dst[i] = c.w[i];
}

// src: `bytes` (multiple of 4) of descriptors on the host; dst: 4-byte aligned device memory; shadow: host copy of what dst holds, or null.
// Returns hipSuccess, or the launch error.
static inline hipError_t upload(void *dst, const void *src, size_t bytes, void *shadow, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    if (shadow && std::memcmp(shadow, src, bytes) == 0) { ctgcn_table_count_(0); return hipSuccess; }
    ctgcn_table_count_(1);
    if (shadow) std::memset(shadow, 0xff, bytes);           // not current until every chunk is queued
    const char *s = (const char *)src;
    for (size_t off = 0; off < bytes; off += CHUNK_BYTES) {
        const size_t nb = bytes - off < (size_t)CHUNK_BYTES ? bytes - off : (size_t)CHUNK_BYTES;
        Chunk c;
        std::memcpy(c.w, s + off, nb);
        hipLaunchKernelGGL(table_write_kernel, dim3(1), dim3(256), 0, st, c, (uint32_t *)((char *)dst + off), (int32_t)((nb + 3) / 4));
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (shadow) std::memcpy(shadow, src, bytes);
    return hipSuccess;
}

}   // namespace ctgcn_table
