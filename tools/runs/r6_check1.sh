#!/bin/bash
# round 6: after the descriptor-table slab, the LayerNorm spreading of the layer kernel and the k-core marking rule
python -m pytest tests/test_gpu_group.py tests/test_gpu_agg_split.py tests/test_gpu_gru.py tests/test_gpu_models.py -q -x 2>&1 | tail -8
python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --no-pmc --detail-file gpurun_out/d_ln.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['also']['kernel_ms_per_step'], d['also'].get('gru_layer_mfma_frac'))"
CTGCN_KCORE_TRACE=1 python tools/kcore_bench.py --snapshots 15 2>&1 | grep "kcore:" | head -60 | awk '{printf "%s/%s ", $5, $8} END {print ""}'
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o kc -- python $R/tools/kcore_bench.py --snapshots 15 > /dev/null 2>&1
python - <<P
import csv,glob
f=glob.glob("/tmp/kc/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    if "kcore" in r["Name"]: print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
P
