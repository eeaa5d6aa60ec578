#!/bin/bash
# round 6: where the exact k-core call spends its time — flags per sweep (CTGCN_KCORE_TRACE=1) and the duration of every dispatch of ONE call
CTGCN_KCORE_TRACE=1 python tools/kcore_bench.py --snapshots 15 2>&1 | grep "kcore:" | head -50 | awk '{printf "%s/%s ", $5, $7} END {print ""}'
R=$PWD; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kc && rocprofv3 --kernel-trace --output-format csv -d /tmp/kc -o kc -- python $R/tools/kcore_bench.py --snapshots 15 > /dev/null 2>&1
python - <<P
import csv,glob
f=glob.glob("/tmp/kc/**/*kernel_trace.csv",recursive=True)
rows=[r for r in csv.DictReader(open(f[0])) if "kcore" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last exact call: from the last hindex_init to the finish kernel after it
starts=[i for i,r in enumerate(rows) if "hindex_init" in r["Kernel_Name"]]
i0=starts[-1]
i1=next(i for i in range(i0,len(rows)) if "finish" in rows[i]["Kernel_Name"])
t0=int(rows[i0]["Start_Timestamp"])
out=[]
for r in rows[i0:i1+1]:
    nm=r["Kernel_Name"]; tag="I" if "init" in nm else "F" if "finish" in nm else "h" if "hub" in nm else "m"
    out.append("%s%.0f" % (tag,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
print("dispatch durations in us (m main sweep, h hub kernel):", " ".join(out))
print("call span %.1f us, sum of kernels %.1f us" % ((int(rows[i1]["End_Timestamp"])-t0)/1e3, sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows[i0:i1+1])/1e3))
P
