O=gpurun_out/r4h; mkdir -p $O; R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 200 python tools/agg_bench.py --split --plan 1 --snapshots 15,7 --iters 5 > $O/agg_powerlaw.txt 2>&1
timeout 200 python tools/agg_bench.py --split --plan 1 --snapshots 15,7 --iters 5 --order core > $O/agg_powerlaw_core_order.txt 2>&1
timeout 200 python tools/agg_bench.py --split --plan 1 --uniform 16 --nodes 1000000 --iters 5 > $O/agg_uniform_1m.txt 2>&1
timeout 200 python tools/agg_bench.py --split --plan 1 --uniform 16 --nodes 4000000 --iters 5 > $O/agg_uniform_4m.txt 2>&1
tail -3 $O/agg_*.txt
cd /tmp; export TMPDIR=/tmp
for pass in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/$O/pmc_$tag -o agg -- python $R/tools/agg_bench.py --split --plan 1 --snapshots 15,7 --iters 2 > $R/$O/pmc_$tag.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_FETCH_core -o agg -- python $R/tools/agg_bench.py --split --plan 1 --snapshots 15,7 --iters 2 --order core > $R/$O/pmc_FETCH_core.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_FETCH_uni4m -o agg -- python $R/tools/agg_bench.py --split --plan 1 --uniform 16 --nodes 4000000 --iters 2 > $R/$O/pmc_FETCH_uni4m.log 2>&1
cd $R
for d in $O/pmc_*/; do echo "== $d"; python tools/pmc_sum.py $d/agg_counter_collection.csv agg_fwd_split32 ; done > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['training_step_ms_per_step'], d['exact_fp32_ms_per_step']); print(json.dumps(d['configs'], indent=1)[:3000]); print(json.dumps(d['cpu_baseline'])[:800])"
