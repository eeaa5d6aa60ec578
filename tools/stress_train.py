#!/usr/bin/env python3
"""Stress: the fused training layer (forward + backward, d = 128, row plan) gives bit-identical gradients run after run.
  python tools/stress_train.py [iters] [nodes]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd.helper import core_adj_from_scipy  # noqa: E402
from ctgcn_amd.layers import CoreDiffusion  # noqa: E402
from ctgcn_amd.synth import dynamic_graph  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
dev = torch.device("cuda:0")
g = dynamic_graph(n, avg_deg=10, snapshots=1, seed=7)[0]
adj, _, _ = core_adj_from_scipy(g, 8, dev)
torch.manual_seed(1)
layer = CoreDiffusion(128, 128).to(dev)
x = torch.randn(n, 128, device=dev, requires_grad=True)
G = torch.randn(n, 128, device=dev)
ref, bad = None, 0
side = torch.cuda.Stream()
for it in range(iters):
    layer.zero_grad(set_to_none=True)
    x.grad = None
    junk = torch.empty((it * 7919) % 30_000_000 + 1, device=dev)
    with torch.cuda.stream(side):                          # a neighbour on another stream: wave timing changes from run to run
        y = torch.randn(4096, 4096, device=dev) @ torch.randn(4096, 4096, device=dev)
    out = layer(x, adj)
    (out * G).sum().backward()
    cur = [out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters() if p.grad is not None]
    del junk, y
    if ref is None:
        ref = cur
    else:
        diff = [i for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
        if diff:
            bad += 1
            print("  mismatch it=%d tensors %s" % (it, diff), flush=True)
torch.cuda.synchronize()
print("fused training layer, %d nodes: %d mismatching runs in %d" % (n, bad, iters), flush=True)
