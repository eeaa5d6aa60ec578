O=gpurun_out/r4at; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_models.py -q -x 2>&1 | grep -v amdgpu.ids | tail -4
for dma in 0 1; do
  echo "== CTGCN_GEMM_DMA=$dma" >> $O/gemm_dma.txt
  CTGCN_GEMM_DMA=$dma timeout 200 python tools/gemm_bench.py --iters 10 2>&1 | grep "split" | cut -c1-150 >> $O/gemm_dma.txt
done
cat $O/gemm_dma.txt
for w in enron-like facebook-like math-like; do
  for dma in 0 1; do
    CTGCN_GEMM_DMA=$dma timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_${w}_$dma.json 2> $O/bench_${w}_$dma.err
    python -c "
import json; d=json.load(open('$O/bench_${w}_$dma.json')); print('$w', 'dma=$dma', d['ms_per_step'], d.get('kernel_ms_per_step_rank0'))" | tee -a $O/gemm_dma.txt
  done
done
