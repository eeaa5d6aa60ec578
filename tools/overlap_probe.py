#!/usr/bin/env python3
"""Experiment: the aggregation (HBM-bound) and the GRU layer kernel (matrix-core / VALU bound) on disjoint CU sets, concurrently.
  python tools/overlap_probe.py [--agg-cus 32]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CoreAdj, _lib, ops  # noqa: E402
from ctgcn_amd.synth import dynamic_graph_device  # noqa: E402


def masked_stream(hip, bits, total=256):
    words = (ctypes.c_uint32 * (total // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), total // 32, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agg-cus", type=int, default=32)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--snapshot", type=int, default=15)
    ap.add_argument("--dedup", type=int, default=1, help="1: both kernels under the graph's row plan (the inference path), 0: every row of H")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lib = _lib.load()
    hip = ctypes.CDLL("libamdhip64.so")
    n = 1_000_000
    rp, col, val = dynamic_graph_device(n, 16, 16, dev, which=[a.snapshot])[a.snapshot]
    adj, _, _ = CoreAdj.from_graph(rp, col, val, max_core=8)
    x = torch.randn(n, 128, device=dev)
    rnn = torch.nn.GRU(128, 128, 1, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(128).to(dev)
    out = torch.empty(n, 128, device=dev)

    plan = adj.row_plan() if a.dedup else None

    def agg():
        return ops.aggregate_split_planes(x, adj, 1, plan)[0]

    def layer(ws):
        ops.gru_layer_presplit(ws, n, adj.K, rnn, norm, out, plan)

    def timeit(fn, iters):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    ws0 = agg()
    t_agg = timeit(agg, a.iters)
    t_layer = timeit(lambda: layer(ws0), a.iters)
    print("full GPU: aggregation %.3f ms, layer kernel %.3f ms, back to back %.3f ms" % (t_agg, t_layer, t_agg + t_layer))
    k = a.agg_cus
    hbm_bits = [i for i in range(256) if (i // 8) % (256 // 8 // (k // 8)) == 0][:k] if k % 8 == 0 else list(range(k))
    # simplest split: bits are spread over the XCDs by the driver; take every (256/k)-th CU for the aggregation
    stride = 256 // k
    hbm_bits = list(range(0, 256, stride))[:k]
    mat_bits = [i for i in range(256) if i not in set(hbm_bits)]
    sH, sM = masked_stream(hip, hbm_bits), masked_stream(hip, mat_bits)
    lib.ctgcn_set_persistent_cus(len(mat_bits))
    with torch.cuda.stream(sH):
        tH = timeit(agg, a.iters)
    with torch.cuda.stream(sM):
        tM = timeit(lambda: layer(ws0), a.iters)
    print("alone on its CUs: aggregation on %d CUs %.3f ms, layer kernel on %d CUs %.3f ms" % (k, tH, len(mat_bits), tM))
    # the layer kernel next to a plain streaming copy on the aggregation's CUs: how sensitive is it to a saturated HBM?
    big_a = torch.empty(1 << 30, dtype=torch.float32, device=dev)
    big_b = torch.empty_like(big_a)
    with torch.cuda.stream(sH):
        t_copy = timeit(lambda: big_b.copy_(big_a), 3)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(sH):
        for _ in range(12):
            big_b.copy_(big_a)
    with torch.cuda.stream(sM):
        ev0.record()
        for _ in range(a.iters):
            layer(ws0)
        ev1.record()
    torch.cuda.synchronize()
    print("streaming copy of 4 GiB on %d CUs: %.3f ms (%.0f GB/s); layer kernel next to it: %.3f ms" % (k, t_copy, 8.59e9 / t_copy / 1e6, ev0.elapsed_time(ev1) / a.iters))
    del big_a, big_b
    # concurrently: N rounds of both
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main_s = torch.cuda.current_stream()
    s.record()
    sH.wait_stream(main_s)
    sM.wait_stream(main_s)
    keep = []
    for _ in range(a.iters):
        with torch.cuda.stream(sH):
            keep.append(agg())
        with torch.cuda.stream(sM):
            layer(ws0)
    main_s.wait_stream(sH)
    main_s.wait_stream(sM)
    e.record()
    torch.cuda.synchronize()
    print("concurrent: %.3f ms per (aggregation + layer) pair  vs %.3f back to back on the full GPU" % (s.elapsed_time(e) / a.iters, t_agg + t_layer))
    lib.ctgcn_set_persistent_cus(0)


if __name__ == "__main__":
    main()
