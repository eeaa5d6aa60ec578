O=gpurun_out/r4ax; mkdir -p $O
for v in saveold savefix; do
  echo "== $v" >> $O/train_layer.txt
  CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 300 python tools/train_layer_bench.py --snapshot 7 --iters 3 2>&1 | grep -v amdgpu.ids | tail -9 >> $O/train_layer.txt
done
cat $O/train_layer.txt
CTGCN_HIP_LIB=$PWD/tools/variants/lib_savefix.so timeout 600 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_models.py -q -x 2>&1 | tail -3
