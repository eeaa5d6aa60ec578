O=gpurun_out/r4t; mkdir -p $O
for v in base sc sc_p; do
  echo "== $v" >> $O/layer_ab.txt
  for s in 3 15; do
    CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 200 python tools/layer_presplit_bench.py --snapshot $s --iters 5 --dedup 1 2>&1 | grep "row plan" >> $O/layer_ab.txt
  done
done
cat $O/layer_ab.txt
for s in 3; do
  CTGCN_HIP_LIB=$PWD/tools/variants/lib_tl.so CTGCN_LAYER_TIMELINE_FILE=$PWD/$O/tl_$s.txt timeout 200 python tools/layer_presplit_bench.py --snapshot $s --iters 2 --dedup 1 2>&1 | grep "row plan" > $O/bench_$s.txt
  cat $O/bench_$s.txt; python tools/layer_timeline.py $O/tl_$s.txt | tee $O/tl_summary_$s.txt
done
CTGCN_HIP_LIB=$PWD/tools/variants/lib_sc.so timeout 600 python -m pytest tests/test_gpu_agg_split.py tests/test_gpu_gru.py tests/test_gpu_group.py -q -x 2>&1 | tail -4
