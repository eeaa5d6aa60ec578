#!/usr/bin/env python3
"""ctgcn_core_aggregate_split_f32 + ctgcn_gru_layer_presplit_f32 on one snapshot of BASELINE config 5 (the two kernels of a width-128
CoreDiffusion layer in inference), timed separately; the command profiled with rocprofv3 --pmc for the layer kernel's counters.
  python tools/layer_presplit_bench.py [--snapshot 15] [--iters 5]"""
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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lib = _lib.load()
    n = a.nodes
    rp, col, val = dynamic_graph_device(n, 16, 16, dev, which=[a.snapshot])[a.snapshot]
    adj, _, _ = CoreAdj.from_graph(rp, col, val, max_core=8)
    x = torch.randn(n, 128, device=dev)
    rnn = torch.nn.GRU(128, 128, 1, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(128).to(dev)
    out = torch.empty(n, 128, device=dev)
    lr = adj.long_rows()
    nl = 0 if lr is None else lr.numel()
    wsb = int(lib.ctgcn_core_aggregate_split_workspace_bytes(n, 128, adj.K, 1, nl))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    bias, b_hn = ops._gru_bias(rnn, 128)
    w_ih, w_hh = rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach().contiguous()
    st = torch.cuda.current_stream().cuda_stream

    def agg():
        _lib.check(lib.ctgcn_core_aggregate_split_f32(n, 128, adj.K, _lib.ptr(adj.row_ptr), _lib.ptr(adj.col), _lib.ptr(adj.val), _lib.ptr(adj.slot),
                                                      _lib.ptr(x), 128, adj.flags | _lib.F_RELU, _lib.ptr(lr), nl, adj.LONG_ROW, 1, _lib.ptr(ws), wsb, st),
                   "ctgcn_core_aggregate_split_f32")

    def layer():
        _lib.check(lib.ctgcn_gru_layer_presplit_f32(n, adj.K, 128, _lib.ptr(ws), _lib.ptr(w_ih), _lib.ptr(w_hh), _lib.ptr(bias), _lib.ptr(b_hn),
                                                    _lib.ptr(norm.weight), _lib.ptr(norm.bias), 1e-5, _lib.ptr(out), 128, st), "ctgcn_gru_layer_presplit_f32")

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

    t_agg, t_layer = timeit(agg), timeit(layer)
    fl = n * (2 * adj.K - 1) * 2.0 * 128 * 384
    print("snapshot %d (K = %d, %d entries): aggregation -> planes %.3f ms | layer kernel on planes %.3f ms = %.1f TF/s fp32-equivalent"
          % (a.snapshot, adj.K, adj.nnz, t_agg, t_layer, fl / t_layer / 1e9))


if __name__ == "__main__":
    main()
