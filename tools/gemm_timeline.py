#!/usr/bin/env python3
"""Per-block timeline of gemm_h2_kernel (diagnostic build of the library with -DCTGCN_GEMM_TIMELINE, see profiles/README.md):
  hipcc ... -DCTGCN_GEMM_TIMELINE -o ctgcn_amd/csrc/libctgcn_hip_timeline.so <sources>
  CTGCN_HIP_LIB=.../libctgcn_hip_timeline.so CTGCN_GEMM_TILE=128 CTGCN_GEMM_TIMELINE_FILE=tl.txt python tools/gemm_bench.py --only 0 --iters 3
  python tools/gemm_timeline.py tl.txt
Columns of the file: block, wall_clock64 (100 MHz) at block start / loop start / loop end / after the stores, XCC_ID<<32 | HW_ID."""
import sys

import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.uint64)
T0, T1, T2, T3 = [d[:, i].astype(np.int64) for i in (1, 2, 3, 4)]
hw = d[:, 5]
ok = T0 > 0
t0 = T0[ok].min()
span = T3[ok].max() - t0
print("blocks %d, kernel span %.1f us" % (ok.sum(), span / 100.0))
for name, v in (("prologue", T1 - T0), ("k loop", T2 - T1), ("epilogue", T3 - T2), ("block life", T3 - T0)):
    v = v[ok] / 100.0
    print("%-10s us: mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90), v.max()))
hwid, xcc = (hw & 0xffffffff).astype(np.int64), (hw >> 32).astype(np.int64) & 0xf
key = xcc * 10000 + ((hwid >> 13) & 7) * 100 + ((hwid >> 12) & 1) * 20 + ((hwid >> 8) & 0xf)
keys = np.unique(key[ok])
busy = []
for k in keys:
    m = ok & (key == k)
    iv = sorted(zip(T0[m], T3[m]))
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy.append((tot + ce - cs) / span)
print("CUs seen %d, a block resident %.1f %% of the span (min %.1f), resident blocks per CU %.2f"
      % (len(keys), 100 * np.mean(busy), 100 * np.min(busy), (T3 - T0)[ok].sum() / span / len(keys)))
