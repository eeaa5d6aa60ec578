#!/usr/bin/env python3
"""rnn_type = 'LSTM' (reference layers.py:27-28): LayerNorm(sum_t LSTM(x)_t) forward and forward + backward on a config-5 sized input
(1 M sequences x 8 steps x 128), the HIP path (ctgcn_lstm_seq_f32 / ctgcn_lstm_seq_bwd_f32 + library GEMMs for the projections) against the
PyTorch-ROCm module evaluated in row chunks (MIOpen; what training used before round 3).
  python tools/lstm_bench.py [--rows 1000000] [--steps 8] [--skip-miopen]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import layers, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--skip-miopen", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rnn = torch.nn.LSTM(128, 128, 1, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(128).to(dev)
    x = torch.relu(torch.randn(a.rows, a.steps, 128, device=dev)).requires_grad_(True)

    def hip_fwd():
        with torch.no_grad():
            return ops.lstm_sequence(rnn, x, norm, True)

    def hip_train():
        x.grad = None
        ops.lstm_sequence(rnn, x, norm, True).square().mean().backward()

    def lib_fwd():
        with torch.no_grad():
            return norm(layers.rnn_over_rows(rnn, x, True))

    def lib_train():
        x.grad = None
        norm(layers.rnn_over_rows(rnn, x, True)).square().mean().backward()

    def timeit(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    print("HIP path:      forward %.2f ms, forward + backward %.2f ms" % (timeit(hip_fwd, 3), timeit(hip_train, 3)), flush=True)
    if not a.skip_miopen:
        print("PyTorch-ROCm:  forward %.2f ms, forward + backward %.2f ms" % (timeit(lib_fwd, 1), timeit(lib_train, 1)), flush=True)
    with torch.no_grad():
        d = (hip_fwd() - lib_fwd()).abs().max().item() if not a.skip_miopen else float("nan")
    print("max |difference| of the two forwards: %.2e" % d)


if __name__ == "__main__":
    main()
