O=gpurun_out/r4an; mkdir -p $O
for st in 2 3 4; do
  CTGCN_STREAMS=$st timeout 900 python tools/stress_group.py 1000 2>&1 | grep -v amdgpu.ids | tail -20 >> $O/stress.txt
done
cat $O/stress.txt
