#!/bin/bash
# Round-5 profile collection on an MI355X box (run from the repo root through gpurun; outputs under gpurun_out/$1).
#   rocprofv3 kernel stats of the config-5 bench and of the four small windows; PMC HBM traffic of the aggregation launches OF THE BENCH
#   COMMAND ITSELF (FETCH_SIZE / WRITE_SIZE in SEPARATE passes, MI355X_MICROARCH.md; VERDICT r4 item 6c: not tools/agg_bench.py).
set -u
OUT=$PWD/gpurun_out/${1:-r5p}
mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- $B > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- $B > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- $B > $OUT/bench_write.json 2> $OUT/bench_write.err
for w in enron-like facebook-like math-like as-like; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -o bench -- python $REPO/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
cd $REPO
python - $OUT <<'P'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for tag, unit in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(out + "/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "agg_fwd_split32" in r["Kernel_Name"] and r["Counter_Name"] == unit] if f else []
    res[tag] = {"launches": len(vals), "mean_KiB": sum(vals) / max(1, len(vals))}
fetch = res["fetch"]["mean_KiB"] * 1024 * 2          # gfx950: FETCH_SIZE tallies 128-byte requests of 16 B/lane streams at 64 B
write = res["write"]["mean_KiB"] * 1024
res["hbm_bytes_per_launch"] = fetch + write
res["fetch_bytes_corrected_x2"] = fetch
res["write_bytes"] = write
json.dump(res, open(out + "/agg_traffic_from_bench.json", "w"), indent=1)
print(json.dumps(res))
P
