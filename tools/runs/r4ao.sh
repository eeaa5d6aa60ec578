O=gpurun_out/r4ao; mkdir -p $O
timeout 900 python tools/stress_train.py 300 30000 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/stress_train.txt
timeout 900 python tools/stress_train.py 60 400000 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a $O/stress_train.txt
