#!/bin/bash
# round 5: is the X stream's DRAM access pattern (128-byte pieces of 1 KB plane rows) what holds gemm_h2_panel_kernel back?
# Same X bytes, same MFMA work, plane rows of 1024 / 512 / 256 / 128 bytes (k = 512 ... 64): at k = 64 a plane is one contiguous stream.
cd "$(dirname "$0")/../.."
S="--shape 435180,512,384 --shape 870360,256,384 --shape 1740720,128,384 --shape 3481440,64,384"
echo "== no epilogue stores (g8)"; CTGCN_HIP_LIB=tools/variants/lib_g8.so python tools/gemm_bench.py --no-lib --iters 10 $S
echo "== no epilogue stores, no MFMA (g12)"; CTGCN_HIP_LIB=tools/variants/lib_g12.so python tools/gemm_bench.py --no-lib --iters 10 $S
echo "== default"; python tools/gemm_bench.py --no-lib --iters 10 $S
