O=gpurun_out/r4al; mkdir -p $O
for st in 2 4 2 4; do
  echo "== CTGCN_STREAMS default overridden? st=$st" >> $O/log.txt
  CTGCN_TEST_STREAMS=$st timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_group.py -q -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -3 >> $O/log.txt
done
cat $O/log.txt
