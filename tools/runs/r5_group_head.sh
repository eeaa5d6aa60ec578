#!/bin/bash
# round 5: the 500-wide first layer of a small window as grouped launches (ops.core_diffusion_wide_group): one part on the main stream, or one
# part per lane (CTGCN_GROUP_HEAD_PARTS, an experiment switch that did not survive: parts never helped / CTGCN_STREAMS), against the per-snapshot
# launches on two lanes (CTGCN_GROUP_HEAD=0).  With the shipped code only the parts=1 rows can be reproduced (CTGCN_GROUP_HEAD=1 forces the grouped form).
# Forward ms per window of the three 'C' shapes.  Output: gpurun_out/r5_group_head.txt
mkdir -p gpurun_out
out=gpurun_out/r5_group_head.txt
: > $out
for w in enron-like math-like as-like; do
  for cfg in 0:1:2 1:1:2 1:2:2 1:3:3 1:4:4; do
    IFS=: read gh parts streams <<< "$cfg"
    echo "== $w CTGCN_GROUP_HEAD=$gh parts=$parts streams=$streams" >> $out
    CTGCN_STREAMS=$streams CTGCN_GROUP_HEAD_PARTS=$parts CTGCN_GROUP_HEAD=$gh timeout 300 python bench.py --workload $w --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>>gpurun_out/r5_group_head.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['value'], j['unit'])" >> $out
  done
done
cat $out
