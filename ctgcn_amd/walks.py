"""Random-walk corpus for the negative-sampling loss (reference preprocessing/random_walk.py:8-69), on the GPU.

random_walk_corpus() returns what the reference stores per snapshot — the symmetric 0/1 co-occurrence matrix
(`walk_spadj`) as a device CSR and the node frequency counts — and negative_table() turns the counts into the
reference's `neg_node_list`.  The walks are drawn by a counter-based RNG (reproducible per seed); the reference uses
numpy's global RNG, so only the deterministic parts are comparable value for value (see tests).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr
from . import ops


class WalkPairs(object):
    """Co-occurrence partners of every node: CSR (row_ptr int32[n+1], col int32[nnz]) on the device.  Stands in for the
    `neighbor_arr = walk_spadj.tolil().rows` lists of the reference loader (helper.py:26-36): len() and [] behave alike."""

    def __init__(self, row_ptr, col):
        self.row_ptr, self.col = row_ptr, col
        self.n = row_ptr.numel() - 1

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        lo, hi = int(self.row_ptr[i]), int(self.row_ptr[i + 1])
        return self.col[lo:hi].tolist()

    @property
    def device(self):
        return self.col.device

    @staticmethod
    def from_lists(rows, device):
        """From the reference's representation: an array/list of per-node python lists."""
        counts = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
        row_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(counts, out=row_ptr[1:])
        col = np.fromiter((c for r in rows for c in r), dtype=np.int32, count=int(row_ptr[-1]))
        return WalkPairs(torch.from_numpy(row_ptr.astype(np.int32)).to(device), torch.from_numpy(col).to(device))

    def to_scipy(self):
        import scipy.sparse as sp
        col = self.col.cpu().numpy()
        return sp.csr_matrix((np.ones(len(col)), col, self.row_ptr.cpu().numpy()), shape=(self.n, self.n))


def random_walk_corpus(row_ptr, col, val, walk_length, walk_time, weighted=True, seed=0, walks_per_round=10):
    """(WalkPairs, freq int64[n]) for one snapshot graph given as device CSR.  Walks are generated `walks_per_round` at a
    time; after each round the pair list is merged into the de-duplicated set (ctgcn_edges_to_csr), so memory stays
    O(n · walks_per_round · L²) instead of O(n · walk_time · L²)."""
    ops._need_cuda(row_ptr, col, val)
    lib = _lib.load()
    dev = row_ptr.device
    n = row_ptr.numel() - 1
    L = int(walk_length)
    per_walk = (L + 1) * L // 2
    freq = torch.zeros(n, dtype=torch.int64, device=dev)
    val = val.to(torch.float32).contiguous()
    cumw = torch.empty_like(val)
    have_src = torch.empty(0, dtype=torch.int32, device=dev)
    have_dst = torch.empty(0, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(lib.ctgcn_row_cumsum_f32(n, ptr(row_ptr), ptr(val), ptr(cumw), ops._stream()), "ctgcn_row_cumsum_f32")
        done = 0
        pr = pc = None
        while done < walk_time:
            w = min(walks_per_round, walk_time - done)
            src = torch.empty(n * w * per_walk, dtype=torch.int32, device=dev)
            dst = torch.empty_like(src)
            check(lib.ctgcn_random_walk_pairs(n, ptr(row_ptr), ptr(col), ptr(cumw), L, w, done, ctypes.c_uint64(seed), 1 if weighted else 0,
                                              ptr(src), ptr(dst), ptr(freq), ops._stream()), "ctgcn_random_walk_pairs")
            pr, pc, _ = ops.edges_to_csr(torch.cat([have_src, src]), torch.cat([have_dst, dst]), None, n)
            # keep one orientation of every unique pair for the next merge
            rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), (pr[1:] - pr[:-1]).long())
            keep = rows < pc
            have_src, have_dst = rows[keep].contiguous(), pc[keep].contiguous()
            done += w
    return WalkPairs(pr, pc), freq


def negative_table(freq):
    """The reference's neg_node_list (random_walk.py:54-60): node i repeated int(((freq_i / total) ** 0.75) / 1e-5) times,
    in node order.  Same float64 arithmetic, vectorised."""
    f = np.asarray(freq.cpu() if isinstance(freq, torch.Tensor) else freq).astype(np.int64)
    tot = f.sum()
    if tot == 0:
        return np.zeros(0, dtype=np.int64)
    rep = (((f / tot) ** 0.75) / 0.00001).astype(np.int64)       # int() truncates toward zero, as astype does for >= 0
    return np.repeat(np.arange(len(f), dtype=np.int64), rep)
