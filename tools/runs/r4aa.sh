O=gpurun_out/r4aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -q -x -s 2>&1 | grep -v amdgpu.ids | tail -25 > $O/tests.txt; cat $O/tests.txt
for g in lib hand; do
  for w in facebook-like; do
    CTGCN_GEMM=$g timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_${w}_$g.json 2> $O/bench_${w}_$g.err
    python -c "
import json; d=json.load(open('$O/bench_${w}_$g.json')); print('$w', '$g', d['ms_per_step'], d.get('kernel_ms_per_step_rank0'))"
  done
done
