O=gpurun_out/r4i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_models.py tests/test_gpu_end_to_end.py -x -q 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
for w in enron-like facebook-like math-like as-like; do
  for g in 1 0; do
    CTGCN_GROUP=$g timeout 200 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/${w}_g$g.json 2> $O/${w}_g$g.err
    python -c "
import json; d=json.load(open('$O/${w}_g$g.json')); r=(d.get('roofline_by_width') or {}).get('128', d['roofline']); print('$w group=$g', d['ms_per_step'], r['frac'], r['avg_launch_ms'], r.get('snapshots_per_launch'), d['kernel_ms_per_step_rank0'])"
  done
done
timeout 300 python bench.py --steps 5 --no-extras --no-cpu-baseline > $O/c5.json 2> $O/c5.err; python -c "
import json; d=json.load(open('$O/c5.json')); print('config5', d['ms_per_step'], d['kernel_ms_per_step_rank0'])"
