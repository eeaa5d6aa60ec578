#!/usr/bin/env python3
"""ctgcn_kcore_i32 on synthetic snapshots of BASELINE config 5 (1 M nodes, average degree 16): peel time, levels, bit-exactness against the oracle's
Batagelj-Zaversnik restatement on a smaller graph.   python tools/kcore_bench.py [--nodes 1000000] [--snapshots 3,15]"""
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
        print("snapshot %d: n = %d, %d entries, level_cap %d: %.3f ms per peel, max core %d, %d distinct core numbers" % (
            t, a.nodes, col.numel(), cap, ms, mk, torch.unique(core).numel()), flush=True)
