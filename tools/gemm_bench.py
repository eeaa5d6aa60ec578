#!/usr/bin/env python3
"""ctgcn_linear_f32 (fp16x2 split GEMM) vs the fp32 library GEMM on the shapes the models produce.
  python tools/gemm_bench.py [--iters 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops  # noqa: E402

SHAPES = [(435_180, 500, 384, "Enron layer-0 GRU projection (87 036 x 5 rows)"), (60_730, 1737, 500, "Facebook-S MLP layer 0"),
          (60_730, 500, 500, "MLP layer 1"), (60_730, 500, 128, "MLP layer 2"), (197_920, 500, 384, "math layer-0 projection (24 740 x 8)")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=int, default=-1, help="index of the single shape to run")
    ap.add_argument("--no-lib", action="store_true")
    ap.add_argument("--shape", action="append", default=[], help="rows,k,n (repeatable): run these instead of the model shapes")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    shapes = SHAPES if a.only < 0 else SHAPES[a.only:a.only + 1]
    if a.shape:
        shapes = [tuple(int(v) for v in sh.split(",")) + ("custom",) for sh in a.shape]
    for rows, k, n, what in shapes:
        x = torch.randn(rows, k, device=dev)
        w = torch.randn(n, k, device=dev) / k ** 0.5
        b = torch.randn(n, device=dev)
        out = torch.empty(rows, n, device=dev)

        def timeit(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / a.iters

        t_split = timeit(lambda: ops.linear_split(x, w, b, out=out))
        # the GEMM kernel alone: both operands prepared once (what the models' cached weights / aggregation-written planes give it)
        from ctgcn_amd import _lib
        from ctgcn_amd._lib import check, ptr
        lib = _lib.load()
        xp, wp = ops._plane_cache.planes(x, lib), ops._plane_cache.packed(w, lib)
        st = ops._stream()
        t_gemm = timeit(lambda: check(lib.ctgcn_linear_packed_f32(rows, n, k, ptr(xp), ptr(wp), ptr(b), 0, ptr(out), n, st), "gemm"))
        t_lib = timeit(lambda: torch.addmm(b, x, w.t(), out=out)) if not a.no_lib else float("nan")
        fl = 2.0 * rows * k * n
        kp = -(-k // 64) * 64
        hbm = rows * kp * 4.0 + rows * n * 4.0
        print("%-52s rows=%d k=%d n=%d: GEMM alone %.3f ms (%.0f TF/s fp32-eq = %.3f of 833; operand + output bytes at %.2f TB/s) | with the split of x %.3f ms | fp32 library %.3f ms (%.0f TF/s)"
              % (what, rows, k, n, t_gemm, fl / t_gemm / 1e9, fl / t_gemm / 1e9 / 833.0, hbm / t_gemm / 1e9, t_split, t_lib, fl / t_lib / 1e9))


if __name__ == "__main__":
    main()
