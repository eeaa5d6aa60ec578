O=gpurun_out/r4ae; mkdir -p $O
for w in enron-like math-like as-like facebook-like; do
  for st in 0 1 2; do
    if [ $st = 0 ]; then unset CTGCN_STREAMS; else export CTGCN_STREAMS=$st; fi
    timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_${w}_$st.json 2> $O/bench_${w}_$st.err
    python -c "
import json; d=json.load(open('$O/bench_${w}_$st.json')); print('$w', 'streams=$st', d['ms_per_step'])"
  done
done
