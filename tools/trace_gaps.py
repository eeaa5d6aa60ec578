#!/usr/bin/env python3
"""Where does a window's wall time go?  rocprofv3 --kernel-trace CSV of a bench.py run -> for the last N steps: wall per step, the union of
the kernel intervals (GPU busy), the serial sum of kernel durations (= busy if nothing overlaps), and the per-kernel serial sums.
  python tools/trace_gaps.py <kernel_trace.csv> <marker kernel substring that occurs once per step> [steps]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marker = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
marks = [i for i, r in enumerate(rows) if marker in r[2]]
marks = marks[-(steps + 1):]
lo, hi = marks[0], marks[-1]
seg = rows[lo:hi]
wall = (rows[hi][0] - rows[lo][0]) / 1e6
busy, cur_s, cur_e = 0, None, None
for s, e, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
serial = sum(e - s for s, e, _ in seg)
n = len(marks) - 1
print("steps %d: wall %.3f ms per step, GPU busy (union) %.3f, serial sum of kernels %.3f, %d launches per step" % (n, wall / n, busy / 1e6 / n, serial / 1e6 / n, len(seg) // n))
acc = collections.Counter()
cnt = collections.Counter()
for s, e, k in seg:
    acc[k[:70]] += e - s
    cnt[k[:70]] += 1
for k, v in acc.most_common(12):
    print("  %8.3f ms  %4d x  %s" % (v / 1e6 / n, cnt[k] // n, k))
