#!/usr/bin/env python3
"""VERDICT r4 item 4, the one structural experiment on the width-128 CoreDiffusion GRU: instead of gru_layer8_h2_kernel (W_ih AND W_hh resident on
the CU, the projection consumed from the accumulators) project the FRESH rows only with the panel GEMM (round 5) into fp32 gate pre-activations and
run the recurrence with only W_hh resident (gru_seq_h2_kernel, the path of the 500-wide first layer) — on snapshots of BASELINE config 5.
  python tools/layer_via_gemm_ab.py [--snapshots 3 15] [--iters 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CoreAdj, _lib, ops  # noqa: E402
from ctgcn_amd._lib import check, ptr  # noqa: E402
from ctgcn_amd.synth import dynamic_graph_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snapshots", type=int, nargs="+", default=[3, 15])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lib = _lib.load()
    n = a.nodes
    x = torch.randn(n, 128, device=dev)
    rnn = torch.nn.GRU(128, 128, 1, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(128).to(dev)
    out_a, out_b = torch.empty(n, 128, device=dev), torch.empty(n, 128, device=dev)
    bias, b_hn = ops._gru_bias(rnn, 128)
    w_hh = rnn.weight_hh_l0.detach().contiguous()
    wp = ops._plane_cache.packed(rnn.weight_ih_l0, lib)

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

    for t in a.snapshots:
        rp, col, val = dynamic_graph_device(n, 16, 16, dev, which=[t])[t]
        adj, _, _ = CoreAdj.from_graph(rp, col, val, max_core=8)
        K = adj.K
        # A: the shipped pair
        plan = adj.row_plan()
        ws, _ = ops.aggregate_split_planes(x, adj, 1, plan)
        ta1 = timeit(lambda: ops.aggregate_split_planes(x, adj, 1, plan, ws=ws))
        ta2 = timeit(lambda: ops.gru_layer_presplit(ws, n, K, rnn, norm, out_a, plan))
        del ws
        # B: aggregation -> compact operand rows (fresh rows only) -> panel GEMM -> recurrence with W_hh resident
        pg = adj.row_plan(adj.PLAN_TILE_GEMM)
        rows = pg["operand_rows"]
        wsg, _ = ops.aggregate_split_planes(x, adj, 384, pg)
        gi = torch.empty(rows * 384, dtype=torch.float32, device=dev)
        st = ops._stream()
        tb1 = timeit(lambda: ops.aggregate_split_planes(x, adj, 384, pg, ws=wsg))
        tb2 = timeit(lambda: check(lib.ctgcn_linear_packed_f32(rows, 384, 128, ptr(wsg), ptr(wp), ptr(bias), 0, ptr(gi), 384, st), "gemm"))
        tb3 = timeit(lambda: check(lib.ctgcn_gru_seq_f32(n, K, 128, ptr(gi), ptr(w_hh), ptr(b_hn), ptr(norm.weight), ptr(norm.bias), float(norm.eps), 1, ptr(out_b), 128,
                                                         None, 2, 0, ptr(pg["order"]), ptr(pg["tile_mask"]), ptr(pg["tile_base"]), st), "gru_seq"))
        err = float((out_a - out_b).abs().max())
        print("snapshot %d (K = %d, %d entries, fresh rows %.3f of %d): layer kernel pair: aggregation %.3f + layer %.3f = %.3f ms | via GEMM: aggregation %.3f + "
              "projection of %d rows %.3f + recurrence %.3f = %.3f ms | max |diff| %.1e" % (t, K, adj.nnz, plan["new_rows"] / float(n * K), n * K, ta1, ta2, ta1 + ta2,
                                                                                         tb1, rows, tb2, tb3, tb1 + tb2 + tb3, err), flush=True)
        del adj, wsg, gi


if __name__ == "__main__":
    main()
