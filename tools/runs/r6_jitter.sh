#!/bin/bash
# Round 6 ("put the jitter build in the loop"): the bit-identity / parity tests, the virtual-rank tests and the stress tools
# against the barrier-jitter build of the CURRENT sources (tools/build_jitter.sh <seed>, ctgcn_amd/csrc/ctgcn_jitter.h: pseudo-random
# wave-dependent delays in front of and behind every __syncthreads()) under four seeds (odd: sparse long delays; even: dense, short ones too).
# Recipe, once per round after the last kernel change:
#     for s in 1 2 3 4; do tools/build_jitter.sh $s; done                      # here (cross-compiles, ~40 s each; the .so files travel with gpurun)
#     gpurun --timeout 2400 -- 'bash tools/runs/r6_jitter.sh 1 2 3 4'          # -> gpurun_out/r6_jitter.txt, copied to profiles/r06_barrier_jitter.txt
mkdir -p gpurun_out
out=gpurun_out/r6_jitter.txt
: > $out
for seed in ${@:-1 2 3 4}; do
  lib=$PWD/tools/variants/lib_jitter$seed.so
  [ -f $lib ] || { echo "missing $lib" >> $out; continue; }
  echo "== CTGCN_HIP_LIB=lib_jitter$seed.so" >> $out
  CTGCN_HIP_LIB=$lib timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_group.py tests/test_gpu_agg_split.py tests/test_gpu_gru.py tests/test_gpu_train_fused.py tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_gpu_virtual_ranks.py -q 2>&1 | grep -v Warn | tail -6 >> $out
  CTGCN_HIP_LIB=$lib timeout 300 python tools/stress_gemm.py --reps 40 2>&1 | grep -v amdgpu.ids | tail -3 >> $out
  CTGCN_HIP_LIB=$lib timeout 300 python tools/stress_train.py 40 30000 2>&1 | grep -v amdgpu.ids | tail -3 >> $out
  CTGCN_HIP_LIB=$lib timeout 300 python tools/stress_group.py 2>&1 | grep -v amdgpu.ids | tail -5 >> $out
  CTGCN_HIP_LIB=$lib timeout 300 python tools/kcore_bench.py --snapshots 3,15 --check 2>&1 | grep -v amdgpu.ids | tail -4 >> $out
done
cat $out
