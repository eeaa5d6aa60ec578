OUT=$PWD/gpurun_out/r4z; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for w in enron-like math-like facebook-like; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -o bench -- python $REPO/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats_$w.csv; head -14 $f | cut -c1-160
done
for pass in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_layer_$tag -o layer -- python $REPO/tools/layer_presplit_bench.py --snapshot 7 --dedup 1 --iters 2 > $OUT/layer_$tag.log 2>&1
  f=$(find $OUT/pmc_layer_$tag -name "*counter_collection.csv" | head -1); python $REPO/tools/pmc_sum.py $f gru_layer8 2>&1 | tail -12
done
cd $REPO; rm -rf $OUT/stats_* 
