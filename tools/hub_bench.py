#!/usr/bin/env python3
"""Cost of one very long row (a hub of --degree neighbours) in the CoreDiffusion aggregation, forward and backward:
the same launch with the hub row on one block (CoreAdj.HUB_SPLIT_MAX = 1) and cut into pieces of 8192 entries (default), and without
the hub at all.  VERDICT r2 item 8: a 200 000-degree hub used to cost 3.8 ms.
  python tools/hub_bench.py [--degree 200000] [--nodes 1000000] [--d 128]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CoreAdj, ops  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=200_000)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=128)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    n = a.nodes
    bg_u, bg_v = rng.integers(1, n, 2 * n), rng.integers(1, n, 2 * n)            # background: average degree 4, no edge touches node 0
    hub_v = rng.choice(np.arange(1, n), a.degree, replace=False)

    def build(with_hub):
        u = np.concatenate([bg_u, np.zeros(a.degree, np.int64)]) if with_hub else bg_u
        v = np.concatenate([bg_v, hub_v]) if with_hub else bg_v
        rp, col, val = ops.edges_to_csr(torch.from_numpy(u.astype(np.int32)).to(dev), torch.from_numpy(v.astype(np.int32)).to(dev), None, n)
        return CoreAdj.from_graph(rp, col, val, max_core=4)[0]

    x = torch.randn(n, a.d, device=dev)
    res = {}
    for label, with_hub, split_max in (("no hub", False, 32), ("hub, one block", True, 1), ("hub, pieces of 8192", True, 32)):
        CoreAdj.HUB_SPLIT_MAX = split_max
        adj = build(with_hub)
        xg = x.clone().requires_grad_(True)
        H = ops.core_aggregate(xg, adj)
        dH = torch.randn_like(H)
        t_f = timeit(lambda: ops._aggregate_fwd(adj, x, True))
        t_b = timeit(lambda: ops._aggregate_bwd(adj, H.detach(), dH, True))
        with torch.no_grad():
            t_s = timeit(lambda: ops.aggregate_split_planes(x, adj, 1, adj.row_plan())) if a.d == 128 else float("nan")
        res[label] = (t_f, t_b, t_s)
        lr = adj.long_rows()
        print("%-22s K = %d, %9d entries, hub rows %d, blocks per hub row %d: forward %.3f ms | backward gather %.3f ms | inference (planes) %.3f ms"
              % (label, adj.K, adj.nnz, 0 if lr is None else lr.numel(), adj.hub_split(), t_f, t_b, t_s), flush=True)
    base = res["no hub"]
    for label in ("hub, one block", "hub, pieces of 8192"):
        print("cost of the %d-entry row, %s: forward %+.3f ms, backward %+.3f ms, inference %+.3f ms"
              % (a.degree, label, res[label][0] - base[0], res[label][1] - base[1], res[label][2] - base[2]))


if __name__ == "__main__":
    main()
