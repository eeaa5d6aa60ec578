#!/usr/bin/env python3
"""The kernels of one width-128 CoreDiffusion layer in TRAINING on a snapshot of BASELINE config 5, timed one by one with HIP events:
forward (aggregation -> planes, layer kernel), backward (recompute, LayerNorm backward, recurrence side, input side, gather).
  python tools/train_layer_bench.py [--snapshot 7] [--iters 3] [--nodes 1000000] [--dedup 1]
Per-kernel bytes are the ALGORITHMIC ones of DESIGN.md §4.4 (what the kernel has to move), rates against them."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CoreAdj, _lib, ops  # noqa: E402
from ctgcn_amd.layers import CoreDiffusion  # noqa: E402
from ctgcn_amd.synth import dynamic_graph_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snapshot", type=int, default=7)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--dedup", type=int, default=1)
    a = ap.parse_args()
    os.environ["CTGCN_DEDUP"] = str(a.dedup)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    _lib.load()
    n = a.nodes
    rp, col, val = dynamic_graph_device(n, 16, 16, dev, which=[a.snapshot])[a.snapshot]
    adj, _, _ = CoreAdj.from_graph(rp, col, val, max_core=8)
    K = adj.K
    layer = CoreDiffusion(128, 128).to(dev)
    x = torch.randn(n, 128, device=dev, requires_grad=True)
    G = torch.randn(n, 128, device=dev)
    plan = adj.row_plan() if a.dedup else None
    fresh = (plan["new_rows"] / float(n * K)) if plan is not None else 1.0
    recs = {}
    ops.set_launch_timer(lambda name, s, e, meta: recs.setdefault(name + ("/save" if meta.get("save") else ""), []).append((s, e)))

    def one():
        layer.zero_grad(set_to_none=True)
        x.grad = None
        out = layer(x, adj)
        (out * G).sum().backward()

    one()
    torch.cuda.synchronize()
    recs.clear()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(a.iters):
        one()
    e0.record()
    torch.cuda.synchronize()
    rs = float(n) * K                              # row-steps
    kb = 1024.0
    bytes_of = {                                   # per call of the whole layer (all chunks)
        "agg_fwd": None,
        "gru_layer": rs * fresh * 516 + n * 512,
        "gru_layer/save": rs * fresh * 516 + rs * 2.0 * kb + n * 512,
        "gru_bwd_rec": rs * 2.0 * kb + n * 512 + rs * fresh * 1.5 * kb,
        "gru_bwd_in": rs * fresh * (1.5 * kb + 516 + 512) + n * 512,
        "agg_bwd": adj.nnz * (4 * 128 + 9) + n * 8 * 128 + 4 * (n + 1),
    }
    print("snapshot %d: n = %d, K = %d, %d stored entries, fresh row-steps %.3f; layer forward + backward %.2f ms per iteration"
          % (a.snapshot, n, K, adj.nnz, fresh, s0.elapsed_time(e0) / a.iters))
    for name in sorted(recs):
        ms = sum(s.elapsed_time(e) for s, e in recs[name]) / a.iters
        b = bytes_of.get(name)
        print("  %-16s %3d launches per iteration, %8.3f ms per iteration%s" % (
            name, len(recs[name]) // a.iters, ms, "" if not b else "  | %.2f GB algorithmic -> %.0f GB/s" % (b / 1e9, b / ms / 1e6)), flush=True)


if __name__ == "__main__":
    main()
