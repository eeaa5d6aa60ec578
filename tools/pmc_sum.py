#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per dispatch of the kernels whose name contains a substring.
  python tools/pmc_sum.py <counter_collection.csv> <kernel substring>"""
import collections
import csv
import sys

acc = collections.defaultdict(float)
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[r["Counter_Name"]] += 1
for c in sorted(acc):
    print("%-40s %.4g  (%d dispatches)" % (c, acc[c] / cnt[c], cnt[c]))
