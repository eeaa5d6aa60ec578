O=gpurun_out/r4az; mkdir -p $O; REPO=$PWD
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -2 $O/bench_full.err
python -c "
import json; d=json.load(open('$O/bench_full.json'))
print('forward', d['ms_per_step'], 'train', d.get('training_step_ms_per_step'), 'exact', d.get('exact_fp32_ms_per_step'), 'value', d['value'], 'copy', d['hbm_copy_GBps_measured']['value'])
print('roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'gru', d['roofline_gru'].get('frac'))
print({k: v.get('ms_per_step') for k, v in d['configs'].items()})
"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/stats -o bench -- python $REPO/bench.py --train --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $REPO/$O/bench_train_under_rocprof.json 2> $REPO/$O/bench_train_under_rocprof.err
f=$(find $REPO/$O/stats -name "*kernel_stats.csv" | head -1); cp $f $REPO/$O/kernel_stats_train.csv; rm -rf $REPO/$O/stats
head -8 $REPO/$O/kernel_stats_train.csv | cut -c1-170
