"""Standalone timing of ctgcn_transpose_bias_f32 (Linear applied to one-hot node features) on the shapes of the bench workloads.
  python tools/transpose_bench.py"""
import torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops
dev = torch.device('cuda:0')
for n, d in ((87036, 500), (1000000, 128), (60730, 500), (24740, 500)):
    lin = torch.nn.Linear(n, d).to(dev)
    for _ in range(3):
        y = ops.linear_of_identity(lin.weight, lin.bias)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        y = ops.linear_of_identity(lin.weight, lin.bias)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print("n=%d d=%d: %.3f ms  %.0f GB/s" % (n, d, ms, 2 * n * d * 4 / ms / 1e6))
