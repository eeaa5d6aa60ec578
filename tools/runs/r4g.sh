mkdir -p gpurun_out/r4g
timeout 300 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_gru.py -x -q 2>&1 | tail -5 > gpurun_out/r4g/tests.txt; cat gpurun_out/r4g/tests.txt
timeout 120 python tools/train_layer_bench.py --snapshot 7 --iters 3 > gpurun_out/r4g/bench7.txt 2>&1; cat gpurun_out/r4g/bench7.txt
timeout 600 python bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r4g/bench.json')); print(d['ms_per_step'], json.dumps(d.get('training_step'))[:900])"
