O=gpurun_out/r4ap; mkdir -p $O
for s in 3 15; do
  CTGCN_HIP_LIB=$PWD/tools/variants/lib_tl.so CTGCN_LAYER_TIMELINE_FILE=$PWD/$O/tl_$s.txt timeout 200 python tools/layer_presplit_bench.py --snapshot $s --iters 2 --dedup 1 2>&1 | grep "row plan" > $O/bench_$s.txt
  cat $O/bench_$s.txt; python tools/layer_timeline.py $O/tl_$s.txt | tee $O/tl_summary_$s.txt
  timeout 200 python tools/layer_presplit_bench.py --snapshot $s --iters 5 --dedup 1 2>&1 | grep "row plan" | tee -a $O/bench_$s.txt
done
