#!/usr/bin/env python3
"""Where do the elementwise copy kernels of a training step come from?  (round 3: 61 ms of a 1 500 ms config-5 step)
  python tools/train_copy_probe.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CTGCN, CoreAdj  # noqa: E402
from ctgcn_amd.synth import dynamic_graph_device  # noqa: E402

dev = torch.device("cuda:0")
n, T = 400_000, 2
graphs = dynamic_graph_device(n, 16, 16, dev, which=[3, 15])
adj = [CoreAdj.from_graph(*graphs[t], max_core=8)[0] for t in (3, 15)]
idx = torch.arange(n, device=dev).repeat(2, 1)
xs = [torch.sparse_coo_tensor(idx, torch.ones(n, device=dev), (n, n)) for _ in range(T)]
torch.manual_seed(0)
model = CTGCN(n, 128, 128, 1, 2, T).to(dev).train()


def step():
    model.zero_grad(set_to_none=True)
    model(xs, adj).square().mean().backward()


step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.events() if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::sum", "aten::cat", "aten::stack") and e.device_time_total > 50]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    st = [s for s in (e.stack or []) if "ctgcn_amd" in s or "bench" in s][:2]
    print("%-18s %8.1f us  shapes %s  %s" % (e.name, e.device_time_total, str(e.input_shapes)[:70], " <- ".join(s.split("/")[-1] for s in st)))
