#!/usr/bin/env python3
"""Randomised check of the HIP training path of the recurrent layers (GRU / LSTM + sum + LayerNorm, layers.py:59-62 / models.py:249-250)
against CPU torch autograd: outputs and every gradient, random shapes.  Not part of the test suite (open-ended).
  python tools/fuzz_rnn_grads.py [--cases 60] [--seed 0]"""
import argparse
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    worst, bad = 0.0, 0
    for case in range(a.cases):
        kind = "LSTM" if rng.random() < 0.4 else "GRU"
        rows = int(rng.choice([1, 3, 15, 16, 17, 31, 33, 64, 100, 1000, 4097, 20000]))
        steps = int(rng.integers(1, 10))
        d_in = int(rng.choice([128, 128, 16, 40, 300]))
        reduce_sum, bias, use_norm = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) > 0), bool(rng.integers(0, 4) > 0)
        torch.manual_seed(case)
        rnn = (torch.nn.LSTM if kind == "LSTM" else torch.nn.GRU)(d_in, 128, 1, bias=bias, batch_first=True)
        norm = torch.nn.LayerNorm(128) if use_norm else None
        if norm is not None:
            with torch.no_grad():
                norm.weight.uniform_(0.5, 1.5)
                norm.bias.uniform_(-0.5, 0.5)
        x = (torch.relu(torch.randn(rows, steps, d_in)) * float(rng.choice([0.3, 1.5, 5.0]))).requires_grad_(True)
        out = rnn(x)[0]
        out = out.sum(1) if reduce_sum else out
        out = norm(out) if norm is not None else out
        G = torch.randn_like(out)
        (out * G).sum().backward()
        rnn_d, norm_d = copy.deepcopy(rnn).to(dev), (copy.deepcopy(norm).to(dev) if norm is not None else None)
        for p in list(rnn_d.parameters()) + (list(norm_d.parameters()) if norm_d is not None else []):
            p.grad = None
        # a gradient that arrives as a strided column (the last CoreDiffusion of a snapshot under the temporal GRU)
        xd = x.detach().to(dev).requires_grad_(True)
        fn = ops.lstm_sequence if kind == "LSTM" else ops.gru_sequence
        got = fn(rnn_d, xd, norm_d, reduce_sum)
        if reduce_sum and rng.random() < 0.5:
            holder = torch.zeros(rows, 3, 128, device=dev)
            wrapped = holder.clone()
            wrapped[:, 1] = got                      # backward hands the layer a strided gradient view
            Gd = torch.zeros(rows, 3, 128, device=dev)
            Gd[:, 1] = G.to(dev)
            (wrapped * Gd).sum().backward()
        else:
            (got * G.to(dev)).sum().backward()
        errs = [((got.detach().cpu() - out.detach()).abs().max() / max(1e-6, float(out.detach().abs().max()))).item()]
        pairs = [(xd.grad, x.grad)] + [(pd.grad, pc.grad) for pd, pc in zip(rnn_d.parameters(), rnn.parameters())]
        if norm is not None:
            pairs += [(norm_d.weight.grad, norm.weight.grad), (norm_d.bias.grad, norm.bias.grad)]
        for gd, gc in pairs:
            errs.append(((gd.cpu() - gc).abs().max() / max(1e-6, float(gc.abs().max()))).item())
        e = max(errs)
        worst = max(worst, e)
        if not np.isfinite(e) or e > 1e-4:
            bad += 1
            print("MISMATCH case %d: %s rows=%d steps=%d d_in=%d reduce=%s bias=%s norm=%s  err %.3e" % (case, kind, rows, steps, d_in, reduce_sum, bias, use_norm, e), flush=True)
        if case % 10 == 0:
            print("case %d (%s rows=%d steps=%d d_in=%d): worst relative error so far %.2e" % (case, kind, rows, steps, d_in, worst), flush=True)
    print("%d cases, %d mismatches, worst error / max|reference| %.2e" % (a.cases, bad, worst))
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
