O=gpurun_out/r4ba; mkdir -p $O
CTGCN_HIP_LIB=$PWD/tools/variants/lib_seqfix.so timeout 900 python -m pytest tests/test_gpu_gru.py tests/test_gpu_agg_split.py tests/test_gpu_gemm.py tests/test_gpu_models.py -q -x 2>&1 | tail -3
for w in enron-like math-like as-like; do
  for v in seqold seqfix; do
    CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_${w}_$v.json 2> $O/bench_${w}_$v.err
    python -c "
import json; d=json.load(open('$O/bench_${w}_$v.json')); print('$w', '$v', d['ms_per_step'], d.get('kernel_ms_per_step_rank0'))" | tee -a $O/seq_ab.txt
  done
done
