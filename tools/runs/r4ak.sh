O=gpurun_out/r4ak; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > $O/tests_full.txt; tail -5 $O/tests_full.txt; grep -n "Error\|assert \|FAILED" $O/tests_full.txt | head -20
