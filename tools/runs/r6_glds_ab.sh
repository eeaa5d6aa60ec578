#!/bin/bash
# round 6: x planes of the row-plan layer kernel by global_load_lds (tools/variants/lib_glds.so = -DCTGCN_X_GLDS=1) against the product
for lib in tools/variants/lib_glds.so ""; do
  echo "== CTGCN_HIP_LIB=$lib"
  CTGCN_HIP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_agg_split.py tests/test_gpu_group.py tests/test_gpu_models.py -q -x 2>&1 | tail -2
  CTGCN_HIP_LIB=$lib timeout 300 python tools/layer_presplit_bench.py --snapshot 3 --iters 5 --dedup 1 2>&1 | tail -2
  CTGCN_HIP_LIB=$lib timeout 300 python tools/layer_presplit_bench.py --snapshot 15 --iters 5 --dedup 1 2>&1 | tail -2
done
