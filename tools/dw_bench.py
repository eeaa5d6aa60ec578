#!/usr/bin/env python
"""Micro-benchmark of the GRU training-side kernels on their own: ctgcn_gru_weight_grad_f32 (dW), ctgcn_gru_input_grad_f32
(dX) and the hipBLASLt GEMMs they replace.  python tools/dw_bench.py [--nodes N] [--steps K] [--iters I]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops  # noqa: E402


def timed(fn, iters):
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
    ap.add_argument("--nodes", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = "cuda:0"
    R = a.nodes * a.steps
    dgi = torch.randn(R, 384, device=dev)
    dghn = torch.randn(R, 128, device=dev)
    x = torch.randn(R, 128, device=dev)
    w = torch.randn(384, 128, device=dev) * 0.1
    out = torch.zeros(384, 128, device=dev)
    dx = torch.empty(R, 128, device=dev)
    part = torch.zeros(ops._DW_PAIRS, 384, 128, device=dev)
    gb = R * 2048 / 1e9
    res = {}
    if a.only in ("", "dw"):
        res["dW_ih kernel"] = timed(lambda: ops._weight_grad(part, dgi, dgi[:, 256:], x, a.steps, False, True), a.iters)
        res["dW_hh kernel (shifted h)"] = timed(lambda: ops._weight_grad(part, dgi, dghn, x, a.steps, True, True), a.iters)
    if a.only in ("", "dx"):
        res["dX kernel"] = timed(lambda: ops._project_grad(dgi, w, dx), a.iters)
    if a.only == "":
        res["hipBLASLt TN (split-K bmm)"] = timed(lambda: ops._accumulate_tn(out, dgi, x), a.iters)
        res["hipBLASLt NN"] = timed(lambda: torch.mm(dgi, w, out=dx), a.iters)
    for k, ms in res.items():
        print("%-32s %8.3f ms   %6.2f TB/s of %.2f GB" % (k, ms, gb / ms, gb))


if __name__ == "__main__":
    main()
