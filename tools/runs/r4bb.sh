O=gpurun_out/r4bb; mkdir -p $O
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > $O/tests_full.txt; tail -4 $O/tests_full.txt; grep -n "FAILED" $O/tests_full.txt | head
timeout 300 python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
