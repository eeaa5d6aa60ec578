#!/usr/bin/env python3
"""Bit-identity of the split GEMM (gemm_h2_panel_kernel) under repetition, alone and with a second stream hammering the memory system
(an aggregation-like gather + another GEMM).  python tools/stress_gemm.py [--reps 200]
Shapes: every column-tile count per wave, few / many panels per block, one and several k stages."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
noise_x = torch.randn(200_000, 500, device=dev)
noise_w = torch.randn(384, 500, device=dev)
noise_idx = torch.randint(0, 200_000, (2_000_000,), device=dev)
bad_total = 0
for rows, k, n in [(1500, 200, 96), (1500, 64, 96), (40_000, 500, 128), (40_000, 500, 256), (40_000, 500, 500), (300_000, 500, 384), (60_730, 1737, 500), (129, 96, 1100)]:
    torch.manual_seed(rows + n)
    x = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    ref = ops.linear_split(x, w, b)
    err = float(((ref.double() - (x.double() @ w.double().t() + b.double())).abs() / ((x.double().abs() @ w.double().abs().t()) + 1e-30)).max())
    bad = 0
    for i in range(a.reps):
        if i % 2:                      # every other repetition with a neighbour on another stream
            with torch.cuda.stream(side):
                ops.linear_split(noise_x, noise_w, None)
                noise_x[noise_idx[:500_000]].sum()
        bad += 0 if torch.equal(ops.linear_split(x, w, b), ref) else 1
    torch.cuda.synchronize()
    bad_total += bad
    print("%7d x %4d x %4d: %d repetitions, %d differ from the first; max err / sum|xw| %.2e" % (rows, k, n, a.reps, bad, err), flush=True)
print("TOTAL mismatches:", bad_total)
sys.exit(1 if bad_total else 0)
