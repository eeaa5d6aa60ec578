#!/usr/bin/env python3
"""Per-stage timeline of gemm_h2_panel_kernel (diagnostic build: GEMM=1 tools/build_variant.sh tl -DCTGCN_GEMM_TIMELINE):
  CTGCN_HIP_LIB=tools/variants/lib_tl.so CTGCN_GEMM_TIMELINE_FILE=tl.txt python tools/gemm_bench.py --only 0 --iters 3 --no-lib
  python tools/gemm_timeline.py tl.txt
File rows: block, wave (0 = wave 0, 1 = wave 7), stage, s_memtime at: stage top | after vmcnt(0) | after barrier + conditional block |
after slab 0 | after the wait for slab 1's W fragments | after slab 1."""
import sys

import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.int64)
blk, wav, stg = d[:, 0], d[:, 1], d[:, 2]
T = d[:, 3:9].astype(np.float64)
names = ["wait vmcnt(0) at the top", "barrier (+ epilogue when the panel changes)", "requests + slab 0", "W requests + wait for slab 1's W", "slab 1"]
nks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for label, sel in (("stages inside a panel", (stg % nks != 0) & (stg > 0)), ("first stage of a panel (epilogue of the one before)", (stg % nks == 0) & (stg > 0))):
    print("%s: %d samples, cycles (s_memtime)" % (label, sel.sum()))
    for i, n in enumerate(names):
        v = (T[sel, i + 1] - T[sel, i])
        print("   %-48s mean %7.0f  p10 %7.0f  p50 %7.0f  p90 %7.0f" % (n, v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
    v = T[sel, 5] - T[sel, 0]
    print("   %-48s mean %7.0f  p50 %7.0f" % ("whole stage", v.mean(), np.percentile(v, 50)))
# stage to stage: top of stage s+1 minus top of stage s for the same (block, wave)
key = blk * 2 + wav
order = np.lexsort((stg, key))
k2, s2, t0 = key[order], stg[order], T[order, 0]
same = (k2[1:] == k2[:-1]) & (s2[1:] == s2[:-1] + 1)
print("stage period (top to top): mean %.0f p50 %.0f cycles over %d pairs" % ((t0[1:] - t0[:-1])[same].mean(), np.percentile((t0[1:] - t0[:-1])[same], 50), same.sum()))
