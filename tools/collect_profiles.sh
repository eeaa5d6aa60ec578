#!/bin/bash
# Round-6 profile collection on an MI355X box (run from the repo root through gpurun; outputs under gpurun_out/$1):
#   rocprofv3 --kernel-trace --stats of the config-5 bench command and of the four small windows (kernel_stats.csv each, copied to
#   profiles/r06_bench_*_kernel_stats.csv).  The PMC traffic of the aggregation launches is measured by bench.py itself since round 6
#   (roofline.traffic / detail: pmc), in separate FETCH_SIZE / WRITE_SIZE passes as MI355X_MICROARCH.md prescribes.
set -u
OUT=$PWD/gpurun_out/${1:-r6p}
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-pmc --detail-file $OUT/detail_under_rocprof.json"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
for w in enron-like facebook-like math-like as-like; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -o bench -- python $REPO/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-pmc --detail-file $OUT/detail_$w.json > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
cd $REPO
for d in stats stats_enron-like stats_facebook-like stats_math-like stats_as-like; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv
done
ls -la $OUT | head -30
head -8 $OUT/stats_kernel_stats.csv | cut -c1-200
