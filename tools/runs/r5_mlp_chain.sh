#!/bin/bash
# round 5: the MLP's hidden activations as operand planes from GEMM to GEMM (CTGCN_MLP_CHAIN) on the Facebook-like CTGCN-S window
mkdir -p gpurun_out
out=gpurun_out/r5_mlp_chain.txt
: > $out
for rep in 1 2; do
for ch in 1 0; do
  echo "== facebook-like CTGCN_MLP_CHAIN=$ch" >> $out
  CTGCN_MLP_CHAIN=$ch timeout 300 python bench.py --workload facebook-like --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>>gpurun_out/r5_mlp_chain.err < /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['kernel_ms_per_step_rank0'])" >> $out
done
done
cat $out
