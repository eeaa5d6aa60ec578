#!/usr/bin/env python3
"""Effect of the hub-row split: 1M-node graph, avg-deg 16, plus ONE vertex adjacent to `--hub` others."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CoreAdj, ops
from ctgcn_amd.synth import powerlaw_edges

ap = argparse.ArgumentParser(); ap.add_argument("--hub", type=int, default=200000); a = ap.parse_args()
dev = torch.device("cuda:0"); n = 1_000_000
u, v = powerlaw_edges(n, 8 * n)
rng = np.random.default_rng(0)
hub = rng.choice(np.arange(1, n), a.hub, replace=False)
u = np.concatenate([u, np.zeros(a.hub, np.int64)]); v = np.concatenate([v, hub])
rp, col, val = ops.edges_to_csr(torch.from_numpy(u.astype(np.int32)).to(dev), torch.from_numpy(v.astype(np.int32)).to(dev), None, n)
adj, core, files = CoreAdj.from_graph(rp, col, val, max_core=8)
x = torch.randn(n, 128, device=dev)
for thr in (2048, 1 << 30):
    CoreAdj.LONG_ROW = thr; adj._long = {}
    for _ in range(2): h = ops.core_aggregate(x, adj)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): h = ops.core_aggregate(x, adj)
    e.record(); torch.cuda.synchronize()
    lr = adj.long_rows()
    print("LONG_ROW=%d hub rows=%s max deg=%d  fwd %.3f ms" % (thr, 0 if lr is None else lr.numel(), int((rp[1:]-rp[:-1]).max()), s.elapsed_time(e) / 5))
