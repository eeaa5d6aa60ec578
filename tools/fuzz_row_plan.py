#!/usr/bin/env python3
"""Randomised cross-check of the inference path (row plan, compact operand rows, hub rows in pieces) against the plan-less path and the
separate fp32-H kernels: every case must agree BIT FOR BIT.  Not part of the test suite (open-ended); run it after touching the
aggregation / GRU kernels.
  python tools/fuzz_row_plan.py [--cases 200] [--seed 0]"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctgcn_amd  # noqa: E402
from ctgcn_amd import CoreAdj  # noqa: E402
from ctgcn_amd.utils import symmetric_csr_from_rows  # noqa: E402


def kcore_list(csr, max_core):
    """the loader's list for one snapshot, built with the library's own device route (bit-exact vs the oracle elsewhere)"""
    from ctgcn_amd.helper import core_adj_from_scipy
    return core_adj_from_scipy(csr, max_core, "cuda:0")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    old_long = CoreAdj.LONG_ROW
    bad = 0
    for case in range(a.cases):
        n = int(rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 200, 1000, 5000, 30000]))
        kind = rng.integers(0, 3)
        d = int(rng.choice([128, 128, 64, 500, 36, 256]))
        CoreAdj.LONG_ROW = int(rng.choice([2048, 2048, 8, 40]))
        try:
            if kind < 2:
                m = int(n * rng.uniform(0.2, 6.0)) + 1
                src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
                if n > 100 and rng.random() < 0.5:            # a few dense nodes
                    hubs = rng.choice(n, 3, replace=False)
                    src = np.concatenate([src, rng.choice(hubs, n // 2)])
                    dst = np.concatenate([dst, rng.integers(0, n, n // 2)])
                csr = symmetric_csr_from_rows(src, dst, rng.integers(1, 4, len(src)) * 0.5, n)
                if csr.nnz == 0:
                    continue
                adj = kcore_list(csr, int(rng.choice([-1, 1, 3, 8])))
                if adj is None:
                    continue
            else:
                K = int(rng.integers(1, 12))
                mats = [sp.random(n, n, density=min(1.0, rng.uniform(0.2, 4.0) / max(n, 1)), random_state=int(rng.integers(1 << 30)), format="csr",
                                  dtype=np.float32) for _ in range(K)]
                adj = CoreAdj.from_matrices(mats, device=dev, self_loop=bool(rng.integers(0, 2)))
            torch.manual_seed(case)
            layer = ctgcn_amd.CoreDiffusion(d, 128).to(dev).eval()
            x = torch.randn(n, d, device=dev) * float(rng.choice([1.0, 30.0, 1e-3]))
            outs = {}
            with torch.no_grad():
                for name, env in (("plan", {"CTGCN_DEDUP": "1", "CTGCN_AGG_SPLIT": "1"}), ("no plan", {"CTGCN_DEDUP": "0", "CTGCN_AGG_SPLIT": "1"}),
                                  ("fp32 H", {"CTGCN_DEDUP": "0", "CTGCN_AGG_SPLIT": "0"})):
                    os.environ.update(env)
                    outs[name] = layer(x, adj)
            torch.cuda.synchronize()
            ok = torch.isfinite(outs["plan"]).all() and torch.equal(outs["plan"], outs["no plan"]) and torch.equal(outs["plan"], outs["fp32 H"])
            if case % 20 == 0:
                print("case %d ok so far (n=%d d=%d K=%d)" % (case, n, d, adj.K), flush=True)
            if not ok:
                bad += 1
                print("MISMATCH case %d: n=%d kind=%d d=%d K=%d long_row=%d nested=%s" % (case, n, kind, d, adj.K, CoreAdj.LONG_ROW, adj.nested), flush=True)
        finally:
            CoreAdj.LONG_ROW = old_long
    print("%d cases, %d mismatches" % (a.cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
