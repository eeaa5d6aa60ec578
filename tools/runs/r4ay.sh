O=gpurun_out/r4ay; mkdir -p $O
for v in savefix bwdbuf; do
  echo "== $v" >> $O/train_layer.txt
  for i in 1 2; do CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 300 python tools/train_layer_bench.py --snapshot 7 --iters 3 2>&1 | grep "gru_bwd_rec\|gru_layer/save\|layer forward" >> $O/train_layer.txt; done
done
cat $O/train_layer.txt
CTGCN_HIP_LIB=$PWD/tools/variants/lib_bwdbuf.so timeout 600 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_models.py tests/test_gpu_gru.py -q -x 2>&1 | tail -3
