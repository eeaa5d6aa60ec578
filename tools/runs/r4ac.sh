O=gpurun_out/r4ac; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_agg_split.py tests/test_gpu_models.py tests/test_gpu_gru.py tests/test_gpu_kernels.py tests/test_gpu_group.py -q -x 2>&1 | grep -v amdgpu.ids | tail -6 > $O/tests.txt; cat $O/tests.txt
