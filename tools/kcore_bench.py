#!/usr/bin/env python3
"""ctgcn_kcore_i32 on synthetic snapshots of BASELINE config 5 (1 M nodes, average degree 16): time per call, max core, and — with --check — the
core numbers against the oracle's Batagelj-Zaversnik restatement (CPU, ~0.2 s per snapshot).  CTGCN_KCORE=peel runs the level-synchronous peel
of rounds 1-4 instead of the h-index sweeps.   python tools/kcore_bench.py [--nodes 1000000] [--snapshots 3,15] [--check]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops  # noqa: E402
from ctgcn_amd.synth import dynamic_graph_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=1_000_000)
ap.add_argument("--snapshots", default="3,15")
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
which = [int(s) for s in a.snapshots.split(",")]
graphs = dynamic_graph_device(a.nodes, 16, 16, dev, which=which)
for t in which:
    rp, col, _ = graphs[t]
    for cap in (-1, 8):
        ops.kcore(rp, col, level_cap=cap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            core, mk = ops.kcore(rp, col, level_cap=cap)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        ok = ""
        if a.check:
            import numpy as np
            import scipy.sparse as sp
            from oracle import oracle as O
            g = sp.csr_matrix((np.ones(col.numel(), dtype=np.float32), col.cpu().numpy(), rp.cpu().numpy()), shape=(a.nodes, a.nodes))
            want = O.core_numbers(g)
            want = want if cap < 0 else np.minimum(want, cap)
            ok = " | equal to the Batagelj-Zaversnik oracle: %s" % bool(np.array_equal(core.cpu().numpy(), want))
        print("snapshot %d: n = %d, %d entries, level_cap %d (%s): %.3f ms per call, max core %d, %d distinct core numbers%s" % (
            t, a.nodes, col.numel(), cap, os.environ.get("CTGCN_KCORE", "default: peel up to cap 16, h-index sweeps beyond"), ms, mk, torch.unique(core).numel(), ok), flush=True)
