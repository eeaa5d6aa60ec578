#!/bin/bash
# round 5: the bit-identity / parity tests and the stress tools against the barrier-jitter build (tools/build_jitter.sh, ctgcn_jitter.h):
# random wave-dependent delays around every __syncthreads().  Output: gpurun_out/r5_jitter.txt
mkdir -p gpurun_out
out=gpurun_out/r5_jitter.txt
: > $out
for seed in ${@:-1}; do
  lib=$PWD/tools/variants/lib_jitter$seed.so
  [ -f $lib ] || { echo "missing $lib" >> $out; continue; }
  echo "== CTGCN_HIP_LIB=lib_jitter$seed.so" >> $out
  CTGCN_HIP_LIB=$lib timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_group.py tests/test_gpu_agg_split.py tests/test_gpu_gru.py tests/test_gpu_train_fused.py tests/test_gpu_kernels.py tests/test_gpu_models.py -q 2>&1 | grep -v Warn | tail -6 >> $out
  CTGCN_HIP_LIB=$lib timeout 300 python tools/stress_gemm.py --reps 60 2>&1 | grep -v amdgpu.ids | tail -18 >> $out
  CTGCN_HIP_LIB=$lib timeout 300 python tools/stress_train.py 60 30000 2>&1 | grep -v amdgpu.ids | tail -4 >> $out
  CTGCN_HIP_LIB=$lib timeout 300 python tools/stress_group.py 2>&1 | grep -v amdgpu.ids | tail -5 >> $out
done
cat $out
