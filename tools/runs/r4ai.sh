O=gpurun_out/r4ai; mkdir -p $O; REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -8 > $O/tests.txt; cat $O/tests.txt
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -3 $O/bench_full.err
python -c "
import json; d=json.load(open('$O/bench_full.json'))
print('forward', d['ms_per_step'], 'train', d.get('training_step_ms_per_step'), 'exact', d.get('exact_fp32_ms_per_step'), 'value', d['value'])
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print({k: v.get('ms_per_step') for k, v in d['configs'].items()})
print('gru', d['roofline_gru']['fused_layers']['frac'] if 'fused_layers' in d['roofline_gru'] else d['roofline_gru'].get('frac'))
"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/stats -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $REPO/$O/bench_under_rocprof.json 2> $REPO/$O/bench_under_rocprof.err
f=$(find $REPO/$O/stats -name "*kernel_stats.csv" | head -1); cp $f $REPO/$O/kernel_stats_fwd.csv; head -8 $f | cut -c1-180
rm -rf $REPO/$O/stats
