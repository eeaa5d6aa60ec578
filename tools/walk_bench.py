#!/usr/bin/env python3
"""Random-walk corpus + negative-sampling draw timings on a config-5 snapshot."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd.synth import dynamic_graph_device
from ctgcn_amd.walks import random_walk_corpus, negative_table
from ctgcn_amd.metrics import NegativeSamplingLoss

dev = torch.device("cuda:0"); n = 1_000_000
rp, col, val = dynamic_graph_device(n, 16, 16, dev, which=[7])[7]
torch.cuda.synchronize(); t0 = time.perf_counter()
pairs, freq = random_walk_corpus(rp, col, val, walk_length=5, walk_time=10, weighted=True, seed=1)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("corpus: 1M nodes x 10 walks x 5 steps -> %d partner entries, %.2f s (%.1f M walk steps/s)" % (pairs.col.numel(), t1 - t0, n * 10 * 5 / (t1 - t0) / 1e6))
table = torch.from_numpy(negative_table(freq).astype(np.int32)).to(dev)
loss = NegativeSamplingLoss([pairs], [table], neg_num=20, Q=20)
batch = torch.randperm(n, device=dev)[:1024]
emb = torch.randn(n, 128, device=dev, requires_grad=True)
for _ in range(3): l = loss([emb, batch])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): l = loss([emb, batch])
torch.cuda.synchronize(); print("NegativeSamplingLoss forward, batch 1024, neg_num 20: %.3f ms (table %d entries)" % ((time.perf_counter() - t0) / 20 * 1e3, table.numel()))
