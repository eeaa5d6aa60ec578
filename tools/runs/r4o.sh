O=gpurun_out/r4o; mkdir -p $O; R=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['training_step_ms_per_step'], d['exact_fp32_ms_per_step']); print({k:(v.get('ms_per_step'), (v.get('roofline_d128') or {}).get('frac')) for k,v in d['configs'].items()}); t=d['training_step']; print(t['kernel_ms_per_step'], t['memory'])"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/bench_under_rocprof.json 2> $R/$O/bench_under_rocprof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_train -o train -- python $R/bench.py --train --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/train_under_rocprof.json 2> $R/$O/train_under_rocprof.err
cd $R; ls $O/stats $O/stats_train
