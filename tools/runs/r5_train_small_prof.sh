#!/bin/bash
# round 5: where a small 'C' window's training step goes (enron-like: 121 ms against a 14.5 ms forward): rocprofv3 kernel stats of the training leg
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_train_small
mkdir -p $O
for w in ${1:-enron-like}; do
  d=$O/$w
  CTGCN_BENCH_REFERENCE_LOSS=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-extras --train-leg --no-cpu-baseline > $d.log 2>$d.err < /dev/null
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $w" >> $O/summary.txt
  if [ -n "$f" ]; then head -40 "$f" | cut -c1-260 >> $O/summary.txt; fi
  python - "$d.log" >> $O/summary.txt <<'P'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        j = json.loads(line)
        print(json.dumps(j.get("training_step"), indent=None)[:3000])
P
  find $d -type f ! -name '*kernel_stats.csv' -delete
done
cat $O/summary.txt
