timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_end_to_end.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
