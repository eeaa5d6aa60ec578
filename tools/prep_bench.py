#!/usr/bin/env python3
"""Preprocessing micro-benchmark on BASELINE config 5 snapshots: GPU ingest (edge rows -> CSR), k-core peel,
level tagging + slot reorder, next to the CPU oracle (single-thread Batagelj-Zaversnik).
  python tools/prep_bench.py [--snapshots 15,7] [--iters 5] [--cpu]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops  # noqa: E402
from ctgcn_amd.core_adj import slot_table  # noqa: E402
from ctgcn_amd.synth import powerlaw_edges, prefix_sizes  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snapshots", default="15,7,0")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = a.nodes
    u, v = powerlaw_edges(n, n * 8)
    sizes = prefix_sizes(len(u), 16)
    ud, vd = torch.from_numpy(u.astype(np.int32)).to(dev), torch.from_numpy(v.astype(np.int32)).to(dev)
    for t in [int(s) for s in a.snapshots.split(",")]:
        m = sizes[t]
        ms_in, (rp, col, val) = timed(lambda: ops.edges_to_csr(ud[:m], vd[:m], None, n), a.iters)
        nnz = col.numel()
        ms_kc, (core, mx) = timed(lambda: ops.kcore(rp, col), a.iters)
        ms_kc8, _ = timed(lambda: ops.kcore(rp, col, level_cap=8), a.iters)

        def tag():
            level, count, wsum = ops.edge_levels(rp, col, val, core, mx + 1)
            table, K, levels, nnzs = slot_table(count.cpu().numpy(), wsum.cpu().numpy(), mx, 8, n)
            return ops.slot_reorder(rp, col, val, level, torch.from_numpy(table).to(dev), K)
        ms_tag, _ = timed(tag, a.iters)
        kc_bytes = 2 * (4 * (n + 1) + 4 * nnz) + 8 * n
        print("t=%2d rows=%8d nnz=%9d maxcore=%3d | ingest %.2f ms (%.0f Mrows/s) | k-core %.2f ms (%.1f Gentries/s, %.1f GB/s min-traffic = %.2f%% of 8 TB/s) | k-core capped at 8 (all max_core=8 needs) %.2f ms | tag+reorder %.2f ms"
              % (t, m, nnz, mx, ms_in, m / ms_in / 1e3, ms_kc, nnz / ms_kc / 1e6, kc_bytes / ms_kc / 1e6, 100 * kc_bytes / ms_kc / 1e6 / 8000, ms_kc8, ms_tag), flush=True)
        if a.cpu:
            from oracle import oracle as O
            import scipy.sparse as sp
            csr = sp.csr_matrix((val.cpu().numpy(), col.cpu().numpy(), rp.cpu().numpy()), shape=(n, n))
            t0 = time.perf_counter()
            ref = O.core_numbers(csr)
            dt = time.perf_counter() - t0
            assert np.array_equal(ref, core.cpu().numpy())
            print("      CPU oracle BZ (1 thread): %.1f ms (%.2f Gentries/s) -> GPU %.0fx" % (dt * 1e3, nnz / dt / 1e9, dt * 1e3 / ms_kc), flush=True)


if __name__ == "__main__":
    main()
