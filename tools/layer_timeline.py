#!/usr/bin/env python3
"""Phase sums of gru_layer8_h2_kernel per (block, wave) from a -DCTGCN_LAYER_TIMELINE build of the library:
  CTGCN_HIP_LIB=.../libctgcn_hip_timeline.so CTGCN_LAYER_TIMELINE_FILE=tl.txt python tools/gru_bench.py --rows 1000000 --iters 1
  python tools/layer_timeline.py tl.txt
Columns: block, wave, wall_clock64 ticks (10 ns) summed over the wave's units for: MFMA stream issue (operand reads + 72 MFMAs +
staging slices), gate math (includes waiting for the last MFMAs), publish (h planes -> LDS), barrier wait, tile ends (LayerNorm +
barrier); units."""
import sys

import numpy as np

d = np.loadtxt(sys.argv[1], dtype=np.float64)
if d.shape[1] == 14 and d[:, 8].sum() > 0:
    # planes + row plan form (the x products of the NEXT unit ride in front of the barrier): ticks for h products issued (gi scaling, operand
    # reads, 36 MFMAs), gate math (+ MFMA drain), publish, x products of the next fresh unit, barrier wait, LayerNorm of the previous tile
    units, fresh = d[:, 8], d[:, 9]
    ok = units > 0
    names = ["h products issued", "gate math (+ MFMA drain)", "publish", "x products issued", "barrier wait", "LayerNorm (prev. tile)", "x staging + request"]
    cols = [2, 3, 4, 5, 6, 7, 10]
    print("waves %d, units per wave %.0f, of them followed by a fresh unit %.3f" % (ok.sum(), units[ok].mean(), fresh[ok].sum() / units[ok].sum()))
    tot = 0.0
    for i, n in enumerate(names):
        per = d[ok, cols[i]].sum() / units[ok].sum() * 10.0
        tot += per
        print("%-28s %8.1f ns per unit" % (n, per))
    print("%-28s %8.1f ns per unit" % ("total", tot))
    for w in range(8):
        m = ok & (d[:, 1] == w)
        print("wave %d: " % w + "  ".join("%7.1f" % (d[m, c].sum() / units[m].sum() * 10.0) for c in cols))
    sys.exit(0)
units = d[:, 7]
ok = units > 0
names = ["MFMA stream issue", "gate math (+ MFMA drain)", "publish", "barrier wait", "tile end (LayerNorm)"]
tot = 0.0
print("waves %d, units per wave %.0f" % (ok.sum(), units[ok].mean()))
for i, n in enumerate(names):
    per = d[ok, 2 + i].sum() / units[ok].sum() * 10.0
    tot += per
    print("%-28s %8.1f ns per unit" % (n, per))
print("%-28s %8.1f ns per unit" % ("total", tot))
for w in range(8):
    m = ok & (d[:, 1] == w)
    print("wave %d: " % w + "  ".join("%7.1f" % (d[m, 2 + i].sum() / units[m].sum() * 10.0) for i in range(5)))
