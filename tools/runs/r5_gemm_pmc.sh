#!/bin/bash
# round 5: SQ counters of gemm_h2_panel_kernel on the Enron projection shape (LIB = variant library or empty)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/gemmpmc_r5; rm -rf $OUT; mkdir -p $OUT
[ -n "$LIB" ] && export CTGCN_HIP_LIB=$PWD/$LIB
python tools/gemm_bench.py --no-lib --iters 20 2>&1 | cut -c54-175
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT -o pmc --output-format csv -- python /root/repo/tools/gemm_bench.py --only 0 --no-lib --iters 5 > $OUT/log.txt 2>&1
cd /root/repo
f=$(find $OUT -name "*counter_collection.csv" | head -1)
python tools/pmc_sum.py $f gemm_h2_panel
k=$(find $OUT -name "*kernel_trace.csv" | head -1)
python - "$k" <<'P'
import csv, sys
d=[float(r["End_Timestamp"])-float(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[1])) if "gemm_h2_panel" in r["Kernel_Name"]]
print("kernel trace: %d dispatches, mean %.1f us" % (len(d), sum(d)/len(d)/1e3))
P
