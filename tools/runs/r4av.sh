O=gpurun_out/r4av; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x 2>&1 | grep -v amdgpu.ids | tail -6
for wide in 0 1; do
  echo "== CTGCN_GEMM_WIDE=$wide" >> $O/gemm_wide.txt
  CTGCN_GEMM_WIDE=$wide timeout 200 python tools/gemm_bench.py --iters 10 2>&1 | grep "split" | cut -c1-150 >> $O/gemm_wide.txt
done
cat $O/gemm_wide.txt
for w in enron-like facebook-like math-like; do
  for wide in 0 1; do
    CTGCN_GEMM_WIDE=$wide timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_${w}_$wide.json 2> $O/bench_${w}_$wide.err
    python -c "
import json; d=json.load(open('$O/bench_${w}_$wide.json')); print('$w', 'wide=$wide', d['ms_per_step'], d.get('kernel_ms_per_step_rank0'))" | tee -a $O/gemm_wide.txt
  done
done
