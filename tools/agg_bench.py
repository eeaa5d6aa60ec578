#!/usr/bin/env python3
"""Micro-benchmark of the aggregation kernels on snapshots of BASELINE config 5 (used for kernel A/B work and as the
command profiled with rocprofv3 --pmc).  Prints per-snapshot kernel time and algorithmic GB/s.
  python tools/agg_bench.py [--snapshots 15,7,0] [--iters 10] [--d 128] [--bwd] [--nodes 1000000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CoreAdj, ops  # noqa: E402
from ctgcn_amd.synth import dynamic_graph_device  # noqa: E402


def alg_bytes(n, nnz, K, d):
    return nnz * (4 * d + 9) + n * K * 4 * d + 4 * (n + 1)


def split_aggregate(x, adj, plan=None):
    from ctgcn_amd import ops
    return ops.aggregate_split_planes(x, adj, 1, plan)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snapshots", default="15")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--max-core", type=int, default=8)
    ap.add_argument("--bwd", action="store_true")
    ap.add_argument("--plan", type=int, default=1, help="with --split: 1 (default) the graph's row plan, 0 every (node, core) row written")
    ap.add_argument("--split", action="store_true", help="ctgcn_core_aggregate_split_f32 (fp16 planes + row scales out) instead of the fp32 H")
    ap.add_argument("--uniform", type=int, default=0, metavar="DEG",
                    help="instead of the power-law window: ONE uniform random graph with this average degree (no hubs: every X row is gathered "
                         "~DEG times at unrelated moments; with --nodes 4000000 X is 2 GB, 8x the 256 MB Infinity Cache — the DRAM-bound bracket of "
                         "the gather, DESIGN.md §4.2)")
    ap.add_argument("--order", default="plan", choices=["plan", "core"],
                    help="with --split --plan 1: `core` = matrix rows relabelled by (capped core number desc, degree desc) before the CSR is built, so that "
                         "hub rows of X are neighbours in memory (the locality experiment of VERDICT r3 item 6)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    which = [int(s) for s in a.snapshots.split(",")]
    if a.uniform:
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        m = a.nodes * a.uniform // 2
        src = torch.randint(0, a.nodes, (m,), device=dev, generator=g, dtype=torch.int32)
        dst = torch.randint(0, a.nodes, (m,), device=dev, generator=g, dtype=torch.int32)
        graphs = {0: ops.edges_to_csr(src, dst, torch.ones(m, device=dev), a.nodes)}
        which = [0]
        del src, dst
    else:
        graphs = dynamic_graph_device(a.nodes, 16, 16, dev, which=which)
    for t in which:
        rp, col, val = graphs[t]
        if a.order == "core":
            # relabel: new id = rank by (capped core desc, degree desc); rebuild the CSR under the new labels (same graph, same results up to
            # the permutation; only the memory order of X's rows changes)
            core, _ = ops.kcore(rp, col, level_cap=a.max_core)
            deg = (rp[1:] - rp[:-1]).long()
            key = core.long() * (int(deg.max().item()) + 1) + deg
            perm = torch.argsort(key, descending=True, stable=True)            # perm[new] = old
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(a.nodes, device=dev)
            rows_old = torch.repeat_interleave(torch.arange(a.nodes, device=dev), deg)
            keep = rows_old < col.long()                                       # one direction of every undirected pair
            rp, col, val = ops.edges_to_csr(inv[rows_old[keep]].to(torch.int32), inv[col.long()[keep]].to(torch.int32), val[keep], a.nodes)
            del core, deg, key, perm, inv, rows_old, keep
        adj, core, files = CoreAdj.from_graph(rp, col, val, max_core=a.max_core)
        x = torch.randn(a.nodes, a.d, device=dev)
        plan = adj.row_plan() if (a.split and a.plan) else None        # the inference path's row plan: repeated rows of H are not written
        run = (lambda: split_aggregate(x, adj, plan)) if a.split else (lambda: ops.core_aggregate(x, adj))
        for _ in range(2):
            h = run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            h = run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        b = alg_bytes(a.nodes, adj.nnz, adj.K, a.d)
        print("t=%2d K=%d nnz=%9d maxcore=%3d  fwd %.3f ms  %.0f GB/s (alg)  %.2f Gedges/s" % (
            t, adj.K, adj.nnz, files, ms, b / ms / 1e6, adj.aggregated_edges / ms / 1e6), flush=True)
        if a.bwd:
            dh = torch.randn_like(h)
            for _ in range(2):
                ops._aggregate_bwd(adj, h, dh, True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(a.iters):
                ops._aggregate_bwd(adj, h, dh, True)
            e.record()
            torch.cuda.synchronize()
            print("      bwd (prep + gather) %.3f ms" % (s.elapsed_time(e) / a.iters), flush=True)
        del h, x, adj


if __name__ == "__main__":
    main()
