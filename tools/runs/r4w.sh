O=gpurun_out/r4w; mkdir -p $O
for v in abl7 abl8 abl9 abl10; do
  echo "== $v" >> $O/layer_abl.txt
  for s in 3 15; do
    CTGCN_HIP_LIB=$PWD/tools/variants/lib_$v.so timeout 200 python tools/layer_presplit_bench.py --snapshot $s --iters 5 --dedup 1 2>&1 | grep "row plan" | sed 's/.*layer kernel on planes/layer kernel/' >> $O/layer_abl.txt
  done
done
cat $O/layer_abl.txt
