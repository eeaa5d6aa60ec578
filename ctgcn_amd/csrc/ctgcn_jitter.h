// Diagnostic builds only (-DCTGCN_JITTER=<seed>, tools/build_jitter.sh): every __syncthreads() of the kernels is wrapped in pseudo-random,
// wave-dependent delays (s_sleep of 0.5 - 8 k cycles in front of and behind the barrier, drawn from the cycle counter, the wave index and
// the seed).  A barrier that is missing somewhere — a wave reading an LDS slot before its producer wrote it, or overwriting one a slow
// wave still reads — shows as a race only when the waves' relative timing happens to open the window (round 4: one forward in 1 200);
// the delays open such windows by orders of magnitude, so that the bit-identity tests and tools/stress_*.py find in minutes what would
// otherwise ship ("no systematic barrier / LDS-hazard check exists").  Results are unchanged by construction: a
// correct kernel gives the same bits with any wave timing.  Not compiled into the product (the macro is undefined there).
#pragma once
#ifdef CTGCN_JITTER
__device__ __forceinline__ void ctgcn_real_syncthreads_() { __syncthreads(); }
__device__ __forceinline__ void ctgcn_jitter_(unsigned salt)
{
    const unsigned t = (unsigned)__builtin_readcyclecounter();
    unsigned h = (t * 2654435761u) ^ ((unsigned)(threadIdx.x >> 6) * 40503u) ^ (blockIdx.x * 9176u) ^ (salt * 69069u + (unsigned)(CTGCN_JITTER));
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    h = (unsigned)__builtin_amdgcn_readfirstlane((int)h);
    switch (h & 15u) {                       // most barriers pass untouched: the kernel's own rhythm stays recognisable
    case 0: __builtin_amdgcn_s_sleep(8); break;
    case 1: __builtin_amdgcn_s_sleep(24); break;
    case 2: __builtin_amdgcn_s_sleep(64); break;
    case 3: __builtin_amdgcn_s_sleep(127); break;
#if (CTGCN_JITTER) % 2 == 0                  // even seeds: half of the barriers are touched, with short delays too
    case 4: __builtin_amdgcn_s_sleep(1); break;
    case 5: __builtin_amdgcn_s_sleep(3); break;
    case 6: __builtin_amdgcn_s_sleep(12); break;
    case 7: __builtin_amdgcn_s_sleep(40); break;
#endif
    default: break;
    }
}
#define __syncthreads() do { ctgcn_jitter_(__LINE__); ctgcn_real_syncthreads_(); ctgcn_jitter_(__LINE__ + 7919u); } while (0)
#endif
