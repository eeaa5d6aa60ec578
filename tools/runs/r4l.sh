O=gpurun_out/r4l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_gru.py -x -q 2>&1 | tail -5 > $O/tests.txt; cat $O/tests.txt
timeout 200 python tools/train_layer_bench.py --snapshot 7 --iters 3 2>&1 | grep -v amdgpu.ids > $O/bench7.txt; cat $O/bench7.txt
