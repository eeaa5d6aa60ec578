#!/bin/bash
# round 5: training step of the small 'C' windows: dW_ih of the 500-wide layer by column slices of the 128-wide weight-gradient kernel
# (CTGCN_WIDE_DW) and the snapshot branches on two streams (CTGCN_TRAIN_STREAMS).  Output: gpurun_out/r5_train_small_ab.txt
mkdir -p gpurun_out
out=gpurun_out/r5_train_small_ab.txt
: > $out
timeout 300 python -m pytest tests/test_gpu_gru.py -x -q -k "gradients or kept" 2>&1 | tail -3 >> $out
for w in enron-like math-like as-like; do
  for cfg in 1:3:0 1:3:1; do
    IFS=: read wd ts kg <<< "$cfg"
    echo "== $w CTGCN_WIDE_DW=$wd CTGCN_TRAIN_STREAMS=$ts CTGCN_KEEP_GI=$kg" >> $out
    CTGCN_BENCH_REFERENCE_LOSS=0 CTGCN_KEEP_GI=$kg CTGCN_WIDE_DW=$wd CTGCN_TRAIN_STREAMS=$ts timeout 200 python bench.py --workload $w --steps 3 --warmup 1 --no-extras --train-leg --no-cpu-baseline 2>>gpurun_out/r5_train_small_ab.err < /dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); t=j['training_step']; print(t['ms_per_step'], t['gradients']['sum_abs'], t['gradients']['max_abs'])" >> $out
  done
done
cat $out
