#!/bin/bash
# round 6: tanh as sign(x)(1 - e)/(1 + e) (product) against 1 - 2/(1 + e^2x) (tools/variants/lib_tanh1.so = -DCTGCN_TANH_V1): parity tests, the
# T = 16 sample's outlier counts, the config-5 kernel times
python -m pytest tests/test_gpu_gru.py tests/test_gpu_models.py tests/test_gpu_agg_split.py tests/test_gpu_group.py tests/test_gpu_train_fused.py -q -x 2>&1 | tail -3
for lib in "" tools/variants/lib_tanh1.so; do
  echo "== CTGCN_HIP_LIB=$lib"
  CTGCN_HIP_LIB=$lib python -m pytest tests/test_gpu_configs.py -q -x -s -k "full_depth or as_c4 or math_c4" 2>&1 | grep -E "vs fp32 oracle|passed|failed" | cut -c1-330
  CTGCN_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-pmc --detail-file gpurun_out/d_tanh.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['also']['kernel_ms_per_step'])"
done
