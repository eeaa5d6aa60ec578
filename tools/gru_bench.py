#!/usr/bin/env python3
"""Micro-benchmark of the fused GRU path: input-projection GEMM (hipBLASLt) and ctgcn_gru_seq_f32, per call.
  python tools/gru_bench.py [--rows 262144] [--steps 8] [--din 128] [--iters 10] [--full-seq]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--din", type=int, default=128)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--full-seq", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rnn = torch.nn.GRU(a.din, 128, 1, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(128).to(dev)
    x = torch.relu(torch.randn(a.rows, a.steps, a.din, device=dev))
    rec = []
    ops.set_launch_timer(lambda name, s, e, meta: rec.append((name, s, e, meta)))
    with torch.no_grad():
        for _ in range(2):
            ops.gru_sequence(rnn, x, norm, not a.full_seq)
        torch.cuda.synchronize()
        rec.clear()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            ops.gru_sequence(rnn, x, norm, not a.full_seq)
        e.record()
        torch.cuda.synchronize()
    total_ms = s.elapsed_time(e) / a.iters
    k_ms = sum(ss.elapsed_time(ee) for nm, ss, ee, _ in rec if nm == "gru_seq") / a.iters
    p_ms = sum(ss.elapsed_time(ee) for nm, ss, ee, _ in rec if nm == "gru_proj") / a.iters
    l_ms = sum(ss.elapsed_time(ee) for nm, ss, ee, _ in rec if nm == "gru_layer") / a.iters
    if l_ms > 0:
        fl = a.rows * (2 * a.steps - 1) * 2.0 * 128 * 384
        print("rows=%d steps=%d: ctgcn_gru_layer_f32 %.3f ms per call = %.1f TF/s fp32-equivalent (%.0f%% of the 833 TF/s fp16x2 bound), "
              "x + out traffic %.0f GB/s" % (a.rows, a.steps, l_ms, fl / l_ms / 1e9, 100 * fl / l_ms / 1e9 / 833.3,
                                             a.rows * (a.steps + (a.steps if a.full_seq else 1)) * 512.0 / l_ms / 1e6))
        return
    fl_rec = a.rows * (a.steps - 1) * 2.0 * 128 * 384
    fl_in = a.rows * a.steps * 2.0 * a.din * 384
    print("rows=%d steps=%d din=%d: total %.3f ms | recurrent kernel %.3f ms = %.1f TF/s (%.0f%% of 157.3) | projection+rest %.3f ms = %.1f TF/s | split-proj kernel %.3f ms (%.0f GB/s)"
          % (a.rows, a.steps, a.din, total_ms, k_ms, fl_rec / k_ms / 1e9, 100 * fl_rec / k_ms / 1e9 / 157.3,
             total_ms - k_ms, fl_in / (total_ms - k_ms) / 1e9, p_ms, a.rows * a.steps * 2048.0 / max(p_ms, 1e-9) / 1e6))


if __name__ == "__main__":
    main()
