O=gpurun_out/r4ab; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_agg_split.py tests/test_gpu_models.py tests/test_gpu_gru.py tests/test_gpu_kernels.py -q -x 2>&1 | grep -v amdgpu.ids | tail -25 > $O/tests.txt; cat $O/tests.txt
for w in enron-like math-like facebook-like as-like; do
  for g in lib hand; do
    CTGCN_GEMM=$g timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_${w}_$g.json 2> $O/bench_${w}_$g.err
    python -c "
import json; d=json.load(open('$O/bench_${w}_$g.json')); print('$w', '$g', d['ms_per_step'], d.get('kernel_ms_per_step_rank0'))"
  done
done
