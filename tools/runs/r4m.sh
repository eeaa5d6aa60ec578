O=gpurun_out/r4m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_agg_split.py tests/test_gpu_models.py tests/test_gpu_configs.py -x -q 2>&1 | tail -8 > $O/tests.txt; cat $O/tests.txt
for w in facebook-like enron-like math-like as-like; do
  for c in 1 0; do
    CTGCN_PLANE_CACHE=$c timeout 200 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/${w}_c$c.json 2> $O/${w}_c$c.err
    python -c "
import json; d=json.load(open('$O/${w}_c$c.json')); print('$w plane_cache=$c', d['ms_per_step'], d['kernel_ms_per_step_rank0'])"
  done
done
