O=gpurun_out/r4ar; mkdir -p $O; REPO=$PWD
timeout 900 python -m pytest tests/test_gpu_gru.py tests/test_gpu_models.py tests/test_gpu_agg_split.py tests/test_gpu_group.py -q -x 2>&1 | grep -v amdgpu.ids | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/stats -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $REPO/$O/bench_under_rocprof.json 2> $REPO/$O/bench_under_rocprof.err
f=$(find $REPO/$O/stats -name "*kernel_stats.csv" | head -1); cp $f $REPO/$O/kernel_stats_fwd.csv; head -6 $f | cut -c1-180
rm -rf $REPO/$O/stats
python -c "
import json; d=json.load(open('$REPO/$O/bench_under_rocprof.json')); print('forward', d['ms_per_step'])"
