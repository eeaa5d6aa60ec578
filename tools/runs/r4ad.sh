OUT=$PWD/gpurun_out/r4ad; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for w in enron-like math-like; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$w -o bench -- python $REPO/bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  f=$(find $OUT/tr_$w -name "*kernel_trace.csv" | head -1)
  python $REPO/tools/trace_gaps.py $f gru_layer8_h2_group_kernel 10 | tee $OUT/gaps_$w.txt
done
rm -rf $OUT/tr_*
