O=gpurun_out/r4bc; mkdir -p $O
for v in cur hp hp14 cur; do
  echo "== $v" >> $O/layer_ab.txt
  for s in 3 15; do
    CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 200 python tools/layer_presplit_bench.py --snapshot $s --iters 5 --dedup 1 2>&1 | grep "row plan" | sed 's/.*layer kernel on planes/layer kernel/' >> $O/layer_ab.txt
  done
done
cat $O/layer_ab.txt
