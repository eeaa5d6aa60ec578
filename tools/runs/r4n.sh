O=gpurun_out/r4n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_models.py tests/test_gpu_group.py -x -q 2>&1 | tail -12 > $O/tests.txt; cat $O/tests.txt
for c in 1 0; do
  CTGCN_MLP_CHAIN=$c timeout 200 python bench.py --workload facebook-like --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/fb_c$c.json 2> $O/fb_c$c.err
  python -c "
import json; d=json.load(open('$O/fb_c$c.json')); print('facebook-like chain=$c', d['ms_per_step'], d['kernel_ms_per_step_rank0'])"
done
