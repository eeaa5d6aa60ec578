#!/bin/bash
# Round-3 profile collection on an MI355X box (run from the repo root through gpurun; outputs under gpurun_out/$1).
#   rocprofv3 kernel stats of the config-5 bench, PMC HBM traffic of the aggregation under the row plan (FETCH_SIZE / WRITE_SIZE in
#   SEPARATE passes, MI355X_MICROARCH.md), SQ counters of the GRU layer kernel under the row plan.
set -u
OUT=$PWD/gpurun_out/${1:-r3p}
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
SN=0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o agg -- python $REPO/tools/agg_bench.py --split --plan 1 --snapshots $SN --iters 2 > $OUT/agg_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o agg -- python $REPO/tools/agg_bench.py --split --plan 1 --snapshots $SN --iters 2 > $OUT/agg_write.log 2>&1
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_layer_$tag -o layer -- python $REPO/tools/layer_presplit_bench.py --snapshot 7 --dedup 1 --iters 2 > $OUT/layer_$tag.log 2>&1
done
cd $REPO
find $OUT -name "*.csv" | head -40 > $OUT/files.txt
python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
for w in enron-like facebook-like math-like as-like; do python bench.py --workload $w --steps 20 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; done
