O=gpurun_out/r4am; mkdir -p $O
for st in 2 1; do
  CTGCN_STREAMS=$st timeout 500 python tools/stress_group.py 300 2>&1 | grep -v amdgpu.ids | tail -20 >> $O/stress.txt
done
echo "== old kernel (variant base), streams 2" >> $O/stress.txt
CTGCN_HIP_LIB=$PWD/tools/variants/lib_oldk.so CTGCN_STREAMS=2 timeout 500 python tools/stress_group.py 300 2>&1 | grep -v amdgpu.ids | tail -12 >> $O/stress.txt
cat $O/stress.txt
