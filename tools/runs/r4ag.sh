O=gpurun_out/r4ag; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_end_to_end.py -q -x 2>&1 | grep -v amdgpu.ids | tail -5
timeout 300 python tools/kcore_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/kcore_new.txt
