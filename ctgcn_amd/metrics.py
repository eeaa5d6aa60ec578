"""NegativeSamplingLoss with the reference's constructor / forward signature (reference metrics.py:18-93).

The reference walks over the batch nodes in Python, drawing positives with random.sample per node (metrics.py:68-84) —
the bottleneck of real training once the model is fast.  Here the draws are one HIP kernel
(ctgcn_neg_sampling_indices); the scores and the BCE terms are the reference's own formulas in torch.
"""
import ctypes
import itertools
import os

import numpy as np
import torch
from torch import nn

from . import _lib, ops
from ._lib import check, ptr
from .walks import WalkPairs

# seed=None streams: a per-process base drawn from OS entropy (the reference calls random.seed() — OS entropy — on every
# forward, metrics.py:69) advanced by a counter, so runs, and the ranks of one job, draw different samples
_seed_base = int.from_bytes(os.urandom(8), "little")
_seed_counter = itertools.count(1)


class NegativeSamplingLoss(nn.Module):
    def __init__(self, node_pair_list, neg_freq_list, neg_num=20, Q=10, seed=None):
        super().__init__()
        self.node_pair_list = node_pair_list        # per snapshot: WalkPairs, or the reference's array of python lists
        self.neg_freq_list = neg_freq_list          # per snapshot: the negative table (list / array / tensor of node ids)
        self.neg_sample_num = neg_num
        self.Q = Q
        self.seed = seed                            # None: a fresh stream per call (the reference reseeds from the OS)
        self.shared_base = None                     # snapshot_parallel.share_loss_seed: one entropy-drawn stream for all ranks of a job
        self._shared_calls = 0
        self._cache = {}

    def _device_inputs(self, i, device):
        hit = self._cache.get((i, str(device)))
        if hit is None:
            pairs = self.node_pair_list[i]
            if not isinstance(pairs, WalkPairs):
                pairs = WalkPairs.from_lists(pairs, device)
            elif pairs.device != device:
                pairs = WalkPairs(pairs.row_ptr.to(device), pairs.col.to(device))
            table = self.neg_freq_list[i]
            table = table if isinstance(table, torch.Tensor) else torch.as_tensor(np.asarray(table, dtype=np.int32))
            hit = self._cache[(i, str(device))] = (pairs, table.to(device=device, dtype=torch.int32).contiguous())
        return hit

    def sample_indices(self, i, batch_indices):
        """(sample_num, node_indices, pos_indices, neg_indices) — metrics.py:62-93 for snapshot i."""
        device = batch_indices.device
        ops._need_cuda(batch_indices)
        pairs, table = self._device_inputs(i, device)
        num = int(self.neg_sample_num)
        batch = batch_indices.to(torch.int64).contiguous()
        deg = (pairs.row_ptr[1:] - pairs.row_ptr[:-1]).long()[batch]
        take = torch.clamp(deg, max=num)
        offsets = torch.cumsum(take, 0) - take
        sample_num = int(take.sum().item())
        if sample_num == 0:
            return 0, None, None, None
        node_idx = torch.empty(sample_num, dtype=torch.int64, device=device)
        pos_idx = torch.empty(sample_num, dtype=torch.int64, device=device)
        neg_idx = torch.empty(num, dtype=torch.int64, device=device)
        scratch = torch.empty(num, dtype=torch.int64, device=device)
        if self.seed is not None:
            seed = self.seed * 1000003 + i
        elif self.shared_base is not None:          # every rank makes the same sequence of calls: the same draws everywhere
            self._shared_calls += 1
            seed = self.shared_base + self._shared_calls * 0x9E3779B97F4A7C15
        else:
            seed = _seed_base + next(_seed_counter) * 0x9E3779B97F4A7C15
        with torch.cuda.device(device):
            check(_lib.load().ctgcn_neg_sampling_indices(batch.numel(), ptr(batch), ptr(pairs.row_ptr), ptr(pairs.col), num, table.numel(),
                                                         ptr(table), ctypes.c_uint64(seed & (2 ** 64 - 1)), ptr(offsets), ptr(node_idx),
                                                         ptr(pos_idx), ptr(neg_idx), ptr(scratch), ops._stream()),
                  "ctgcn_neg_sampling_indices")
        dt = batch_indices.dtype
        return sample_num, node_idx.to(dt), pos_idx.to(dt), neg_idx.to(dt)

    def forward(self, input_list):
        assert len(input_list) == 2
        node_embedding, batch_indices = input_list[0], input_list[1]
        if not isinstance(node_embedding, list) and node_embedding.dim() == 2:
            node_embedding = [node_embedding]
        bce = nn.BCEWithLogitsLoss()
        loss = torch.zeros(1, device=batch_indices.device)
        for i in range(len(node_embedding)):
            emb = node_embedding[i]
            sample_num, node_idx, pos_idx, neg_idx = self.sample_indices(i, batch_indices)
            if sample_num == 0:
                continue
            pos_score = torch.sum(emb[node_idx].mul(emb[pos_idx]), dim=1)
            neg_score = torch.sum(emb[node_idx].matmul(torch.transpose(emb[neg_idx], 1, 0)), dim=1)
            loss = loss + bce(pos_score, torch.ones_like(pos_score)) + self.Q * bce(neg_score, torch.zeros_like(neg_score))
        return loss
