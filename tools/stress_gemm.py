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
# the chained form (round 5: the output leaves as the next layer's operand planes; row maxima exchanged between the eight waves through LDS)
for rows, k, n in [(60_730, 1737, 500), (60_730, 500, 500), (1500, 64, 96), (40_000, 500, 128), (300_000, 200, 384)]:
    torch.manual_seed(rows + n + 1)
    x = torch.randn(rows, k, device=dev)
    w = torch.randn(n, k, device=dev) / k ** 0.5
    b = torch.randn(n, device=dev)
    ref = ops.linear_split(x, w, b, selu=True, planes_out=True).buf.clone()
    used = rows * (-(-n // 64) * 64) * 4 + rows * 4
    bad = 0
    for i in range(a.reps):
        if i % 2:
            with torch.cuda.stream(side):
                ops.linear_split(noise_x, noise_w, None)
                noise_x[noise_idx[:500_000]].sum()
        bad += 0 if torch.equal(ops.linear_split(x, w, b, selu=True, planes_out=True).buf[:used], ref[:used]) else 1
    torch.cuda.synchronize()
    bad_total += bad
    print("chained %7d x %4d x %4d: %d repetitions, %d differ from the first" % (rows, k, n, a.reps, bad), flush=True)
# the grouped launches of the 500-wide first layer (round 5: one panel GEMM over the rows of all snapshots, per-snapshot weights by a
# panel -> snapshot table, the scales of the leaving and the entering snapshot in two LDS slots) against the per-snapshot kernels
from ctgcn_amd.helper import core_adj_from_scipy  # noqa: E402
from ctgcn_amd.layers import CoreDiffusion  # noqa: E402
from ctgcn_amd.synth import dynamic_graph  # noqa: E402
for n_nodes, T, d in [(3001, 5, 500), (3001, 5, 64), (20_000, 4, 200)]:
    adjs = [core_adj_from_scipy(g, 6, dev)[0] for g in dynamic_graph(n_nodes, avg_deg=6, snapshots=T, seed=3)]
    torch.manual_seed(n_nodes + d)
    mods = [CoreDiffusion(d, 128, a_.K).to(dev).eval() for a_ in adjs]
    xs = [torch.randn(n_nodes, d, device=dev) for _ in range(T)]
    rnns, norms = [m.rnn for m in mods], [m.norm for m in mods]
    with torch.no_grad():
        assert ops.core_diffusion_wide_group_ok(xs, adjs, rnns, norms)
        ref = [ops.core_diffusion_split(xs[t], adjs[t], rnns[t], norms[t]) for t in range(T)]
        bad = 0
        for i in range(a.reps):
            if i % 2:
                with torch.cuda.stream(side):
                    ops.linear_split(noise_x, noise_w, None)
                    noise_x[noise_idx[:500_000]].sum()
            junk = torch.empty((i * 7919) % 5_000_000 + 1, device=dev)      # perturb the allocator: the shared planes move
            outs = [torch.empty(n_nodes, 128, device=dev) for _ in range(T)]
            ops.core_diffusion_wide_group(xs, adjs, rnns, norms, outs)
            bad += 0 if all(torch.equal(o, r) for o, r in zip(outs, ref)) else 1
            del junk
    torch.cuda.synchronize()
    bad_total += bad
    print("grouped first layer, %d nodes x %d snapshots, d = %d: %d repetitions, %d differ from the per-snapshot kernels" % (n_nodes, T, d, a.reps, bad), flush=True)
print("TOTAL mismatches:", bad_total)
sys.exit(1 if bad_total else 0)
