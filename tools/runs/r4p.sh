O=gpurun_out/r4p; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_nccl.py -x -q 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt
CTGCN_FORCE_DIST=1 timeout 400 python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/forced_dist.json 2> $O/forced_dist.err; tail -2 $O/forced_dist.err
python -c "
import json; d=json.load(open('$O/forced_dist.json')); print(d['ms_per_step'], d['per_rank_ms'])"
