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

# round 5: a whole training step of a small one-hot 'C' window (500-wide first layer: kept gi, dW_ih by column slices, the snapshot branches
# on three HIP streams) — gradients bit-identical run after run and to the single-stream step
from ctgcn_amd import CTGCN  # noqa: E402
n2, T = 6000, 6
adjs = [core_adj_from_scipy(g_, 6, dev)[0] for g_ in dynamic_graph(n2, avg_deg=8, snapshots=T, seed=11)]
torch.manual_seed(2)
model = CTGCN(n2, 500, 128, 1, 2, T, model_type="C", trans_activate_type="L").to(dev).train()
idx = torch.arange(n2, device=dev)
eye = torch.sparse_coo_tensor(torch.stack([idx, idx]), torch.ones(n2, device=dev), (n2, n2)).coalesce()
xs = [eye for _ in range(T)]
G2 = torch.randn(T, n2, 128, device=dev)
ref, bad = None, 0
for it in range(max(10, iters // 2)):
    os.environ["CTGCN_TRAIN_STREAMS"] = "1" if it % 4 == 3 else "3"
    for p in model.parameters():
        p.grad = None
    junk = torch.empty((it * 7919) % 30_000_000 + 1, device=dev)
    out = model(xs, adjs)
    (out * G2).sum().backward()
    torch.cuda.synchronize()
    cur = [out.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
    del junk, out
    if ref is None:
        ref = cur
    else:
        diff = [i for i, (a, b) in enumerate(zip(cur, ref)) if not torch.equal(a, b)]
        if diff:
            bad += 1
            print("  window mismatch it=%d tensors %s" % (it, diff[:8]), flush=True)
print("training step of a %d-node x %d-snapshot one-hot window on 3 streams / 1 stream: %d mismatching runs in %d" % (n2, T, bad, max(10, iters // 2)), flush=True)
