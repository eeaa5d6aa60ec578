#!/bin/bash
# round 5: gemm_h2_panel_kernel ablation builds (WRONG results) + non-temporal X loads off, Enron / math / Facebook layer-0 shapes
cd "$(dirname "$0")/../.."
echo "== default"; python tools/gemm_bench.py --no-lib --iters 20
for v in g1 g2 g4 g8 gnt0; do echo "== $v"; CTGCN_HIP_LIB=tools/variants/lib_$v.so python tools/gemm_bench.py --no-lib --iters 20; done
