O=gpurun_out/r4as; mkdir -p $O
for v in g0 g1 g2 g3 g4 g5; do
  echo "== $v" >> $O/gemm_abl.txt
  CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 200 python tools/gemm_bench.py --iters 10 2>&1 | grep "split" | cut -c1-150 >> $O/gemm_abl.txt
done
cat $O/gemm_abl.txt
