#!/bin/bash
# round 5: per-kernel time of a math-like / enron-like window with the grouped first layer and with the per-snapshot launches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_head_prof
mkdir -p $O
for w in math-like enron-like; do
  for gh in 1 0; do
    d=$O/${w}_gh$gh
    CTGCN_GROUP_HEAD=$gh timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/bench.py --workload $w --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $d.log 2>&1 < /dev/null
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    echo "== $w gh=$gh" >> $O/summary.txt
    if [ -n "$f" ]; then head -14 "$f" | cut -c1-220 >> $O/summary.txt; fi
    find $d -type f ! -name '*kernel_stats.csv' -delete
  done
done
cat $O/summary.txt
