O=gpurun_out/r4ah; mkdir -p $O
echo "== old (one launch per level 0..max, host reads back every 16)" > $O/kcore_ab.txt
CTGCN_HIP_LIB=$PWD/tools/variants/lib_kc_old.so timeout 300 python tools/kcore_bench.py 2>&1 | grep snapshot >> $O/kcore_ab.txt
echo "== new (level chosen on the device, batches of 32)" >> $O/kcore_ab.txt
timeout 300 python tools/kcore_bench.py 2>&1 | grep snapshot >> $O/kcore_ab.txt
cat $O/kcore_ab.txt
