O=gpurun_out/r4j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_agg_split.py tests/test_gpu_group.py tests/test_gpu_train_fused.py tests/test_gpu_models.py -x -q 2>&1 | tail -15 > $O/tests.txt; cat $O/tests.txt
for p in 1 0; do echo "== CTGCN_LAYER_PAIR=$p"; CTGCN_LAYER_PAIR=$p timeout 300 python tools/layer_presplit_bench.py --snapshot 3 --iters 5 --dedup 1; CTGCN_LAYER_PAIR=$p timeout 300 python tools/layer_presplit_bench.py --snapshot 15 --iters 5 --dedup 1; done > $O/layer_ab.txt 2>&1; cat $O/layer_ab.txt
