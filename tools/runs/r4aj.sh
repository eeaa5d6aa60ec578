O=gpurun_out/r4aj; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_nccl.py -q -x 2>&1 | grep -v amdgpu.ids | tail -60 > $O/nccl.txt; head -70 $O/nccl.txt
