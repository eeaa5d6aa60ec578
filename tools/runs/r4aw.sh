O=gpurun_out/r4aw; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gru.py tests/test_gpu_group.py tests/test_gpu_models.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
for w in enron-like math-like as-like facebook-like; do
  for bc in 0 1; do
    CTGCN_BIAS_CACHE=$bc timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_${w}_$bc.json 2> $O/bench_${w}_$bc.err
    python -c "
import json; d=json.load(open('$O/bench_${w}_$bc.json')); print('$w', 'bias_cache=$bc', d['ms_per_step'])" | tee -a $O/bias_cache.txt
  done
done
