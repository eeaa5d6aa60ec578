mkdir -p gpurun_out/r4f
for m in 0 1 2 4 6 8 15; do echo "== ablate $m"; CTGCN_BWD_ABLATE=$m timeout 120 python tools/train_layer_bench.py --snapshot 7 --iters 2 2>&1 | grep "gru_bwd"; done > gpurun_out/r4f/ablate.txt 2>&1
cat gpurun_out/r4f/ablate.txt
