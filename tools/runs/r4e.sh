mkdir -p gpurun_out/r4e; R=$PWD
python tools/train_layer_bench.py --snapshot 7 --iters 3 > gpurun_out/r4e/bench7.txt 2>&1; cat gpurun_out/r4e/bench7.txt
cd /tmp; export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/r4e/pmc_$tag -o t -- python $R/tools/train_layer_bench.py --snapshot 7 --iters 1 > $R/gpurun_out/r4e/pmc_$tag.log 2>&1
done
cd $R; find gpurun_out/r4e -name "*counter_collection.csv" | head
