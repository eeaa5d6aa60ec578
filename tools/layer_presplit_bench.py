#!/usr/bin/env python3
"""ctgcn_core_aggregate_split_f32 + ctgcn_gru_layer_presplit_f32 on one snapshot of BASELINE config 5 (the two kernels of a width-128
CoreDiffusion layer in inference), timed separately; the command profiled with rocprofv3 --pmc for the layer kernel's counters.
  python tools/layer_presplit_bench.py [--snapshot 15 | --snapshot -1 (all 16)] [--iters 5] [--dedup 0|1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CoreAdj, _lib, ops  # noqa: E402
from ctgcn_amd.synth import dynamic_graph_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snapshot", type=int, default=15)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--dedup", type=int, default=-1, help="1 / 0: with / without the graph's row plan (repeated rows of H skipped); -1: both, compared bit for bit")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lib = _lib.load()
    n = a.nodes
    x = torch.randn(n, 128, device=dev)
    rnn = torch.nn.GRU(128, 128, 1, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(128).to(dev)
    out = torch.empty(n, 128, device=dev)

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / a.iters

    snaps = [a.snapshot] if a.snapshot >= 0 else list(range(16))
    tot = {}
    for t in snaps:
        rp, col, val = dynamic_graph_device(n, 16, 16, dev, which=[t])[t]
        adj, _, _ = CoreAdj.from_graph(rp, col, val, max_core=8)
        fl = n * (2 * adj.K - 1) * 2.0 * 128 * 384
        ref = None
        for dedup in ([False, True] if a.dedup < 0 else [bool(a.dedup)]):
            plan = adj.row_plan() if dedup else None
            ws, _ = ops.aggregate_split_planes(x, adj, 1, plan)
            t_agg = timeit(lambda: ops.aggregate_split_planes(x, adj, 1, plan, ws=ws))
            t_layer = timeit(lambda: ops.gru_layer_presplit(ws, n, adj.K, rnn, norm, out, plan))
            same = ""
            if ref is None:
                ref = out.clone()
            else:
                same = " | bit-identical to the plain path: %s" % bool(torch.equal(ref, out))
            kept = (plan["new_rows"] / float(n * adj.K)) if plan is not None else 1.0
            print("snapshot %d (K = %d, %d entries) %s: aggregation -> planes %.3f ms | layer kernel on planes %.3f ms = %.1f TF/s fp32-equivalent "
                  "(reference flops) | rows of H written %.3f%s" % (t, adj.K, adj.nnz, "row plan" if dedup else "plain   ", t_agg, t_layer,
                                                                    fl / t_layer / 1e9, kept, same), flush=True)
            k = tot.setdefault(dedup, [0.0, 0.0])
            k[0] += t_agg
            k[1] += t_layer
        del adj, ws
    if len(snaps) > 1:
        for dedup, (ta, tl) in tot.items():
            print("window (%d snapshots, one layer each) %s: aggregation %.2f ms, layer kernel %.2f ms" % (len(snaps), "row plan" if dedup else "plain   ", ta, tl))


if __name__ == "__main__":
    main()
