O=gpurun_out/r4aq; mkdir -p $O
for v in tg1 tg4 tg8 tg32; do
  echo "== $v" >> $O/layer_ab.txt
  for s in 3 15; do
    CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 200 python tools/layer_presplit_bench.py --snapshot $s --iters 5 --dedup 1 2>&1 | grep "row plan" | sed 's/.*layer kernel on planes/layer kernel/' >> $O/layer_ab.txt
  done
done
cat $O/layer_ab.txt
CTGCN_HIP_LIB=$PWD/tools/variants/lib_tg8.so timeout 900 python -m pytest tests/test_gpu_agg_split.py tests/test_gpu_gru.py tests/test_gpu_group.py tests/test_gpu_train_fused.py tests/test_gpu_models.py -q -x 2>&1 | tail -3
