"""oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the reference's CTGCN hot path, used only as the *checker*:
  tests/ , __graft_entry__.smoke() , bench.py's cpu_baseline leg.
Nothing under ctgcn_amd/ imports this module (tests/test_host_logic.py::test_product_never_imports_the_oracle enforces it).

Parity status: PINNED against vectors produced by running the reference in the build
container (tests/golden/make_golden.py -> tests/golden/*.npz; checked by
tests/test_oracle_golden.py).  The reference's own test-suite holds no vectors for this path.

Each function cites the reference lines it restates (paths relative to the reference root).
Heavy loops live in ctgcn_oracle.c (same directory), loaded through ctypes.
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libctgcn_oracle.so")
_lib = None


def build(force=False):
    """gcc -O2 -fopenmp the C restatement next to this file (no reference sources involved)."""
    src = os.path.join(_HERE, "ctgcn_oracle.c")
    import hashlib
    with open(src, "rb") as fh:
        digest = hashlib.sha256(fh.read()).hexdigest()
    stamp = _SO + ".srchash"
    current = os.path.exists(_SO) and os.path.exists(stamp) and open(stamp).read().strip() == digest
    if force or not current:           # content hash, not mtimes (a copied snapshot has arbitrary mtimes)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", _SO, src])
        with open(stamp, "w") as fh:
            fh.write(digest + "\n")
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.oracle_kcore_bz.restype = ctypes.c_int
        L.oracle_spmm_csr_f32.restype = ctypes.c_int
        L.oracle_core_aggregate_f32.restype = ctypes.c_int
        L.oracle_core_aggregate_bwd_f32.restype = ctypes.c_int
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _canon(m, dtype=np.float32):
    m = sp.csr_matrix(m)
    m.sum_duplicates()
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(dtype)


# ------------------------------------------------------------------------------- graph input
def adjacency_from_edge_rows(src, dst, w, n):
    """Undirected simple weighted graph from edge-list rows, as utils.py:23-30 (get_nx_graph) and
    utils.py:35-58 (get_sp_adj_mat) both build it: every row sets weight(u,v)=weight(v,u)=w, so the
    LAST row naming an unordered pair wins; self loops are dropped.  Returns scipy CSR float64,
    symmetric, zero diagonal, rows/cols in node-list order (structure_generation.py:52-53)."""
    last = {}
    for s, t, ww in zip(np.asarray(src).tolist(), np.asarray(dst).tolist(), np.asarray(w).tolist()):
        if s == t:
            continue
        last[(s, t) if s < t else (t, s)] = ww
    if not last:
        return sp.csr_matrix((n, n), dtype=np.float64)
    uv = np.array(list(last.keys()), dtype=np.int64)
    ww = np.array(list(last.values()), dtype=np.float64)
    m = sp.coo_matrix((np.concatenate([ww, ww]), (np.concatenate([uv[:, 0], uv[:, 1]]),
                                                  np.concatenate([uv[:, 1], uv[:, 0]]))), shape=(n, n)).tocsr()
    m.sort_indices()
    return m


# ---------------------------------------------------------------------------------- k-core
def core_numbers(adj):
    """structure_generation.py:35 (networkx.core_number on the simple undirected graph)."""
    indptr, indices, _ = _canon(adj)
    n = adj.shape[0]
    core = np.zeros(n, dtype=np.int32)
    rc = lib().oracle_kcore_bz(ctypes.c_int64(n), _p(indptr), _p(indices), _p(core))
    assert rc == 0
    return core


def kcore_matrices(adj, core=None):
    """structure_generation.py:47-56: for k = 1..max_core the subgraph induced by {v: core[v] >= k}
    (networkx.k_core with the precomputed core numbers), emitted over the FULL node list, i.e. an
    n x n CSR holding every edge whose two endpoints both have core >= k, weights kept.
    Returns [A_1, ..., A_maxcore] (empty list when the graph has no edges)."""
    adj = sp.csr_matrix(adj)
    if core is None:
        core = core_numbers(adj)
    coo = adj.tocoo()
    lvl = np.minimum(core[coo.row], core[coo.col])
    out = []
    for k in range(1, int(core.max(initial=0)) + 1):
        keep = lvl >= k
        m = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=adj.shape)
        m.sort_indices()
        out.append(m)
    return out


def core_file_names(max_core):
    """utils.py:142-148 get_format_str as used at structure_generation.py:46,54: zero padded to the
    decimal width of max_core so that sorted() is numeric order."""
    width = len(str(max_core))
    return [str(k).zfill(width) + ".npz" for k in range(1, max_core + 1)]


# ----------------------------------------------------------------------------------- loader
def core_adj_list(per_snapshot_matrices, start_idx, duration, max_time_num, max_core=-1):
    """helper.py:51-82 get_core_adj_list, on in-memory [A_1..A_Kt] lists instead of .npz files.

    quirk A (helper.py:61-62): max_core == -1 is replaced by the FIRST visited snapshot's file count
            and stays at that value for every later snapshot.
    helper.py:63-64: keep files 1..max_core, visit them from the largest kept k downwards.
    helper.py:71-72: first visited matrix gets + I.
    quirk B (helper.py:74-76): a later matrix is dropped when (A_k - A_{k+1}).sum() == 0; otherwise the
            FULL A_k is used (not the difference, no + I).
    utils.py:89-95: values become float32.
    Returns list[T] of list[K_t] of canonical scipy CSR float32 (what coalescing the COO tensor gives)."""
    assert start_idx < len(per_snapshot_matrices)
    out = []
    for t in range(start_idx, min(start_idx + duration, max_time_num)):
        mats = list(per_snapshot_matrices[t])
        if max_core == -1:
            max_core = len(mats)
        mats = mats[:max_core][::-1]
        cur, prev = [], None
        for j, a in enumerate(mats):
            a = sp.csr_matrix(a)
            if j == 0:
                use = a + sp.eye(a.shape[0])
            else:
                if (a - prev).sum() == 0:
                    prev = a
                    continue
                use = a
            prev = a
            m = sp.csr_matrix(use).astype(np.float32)
            m.sum_duplicates()
            m.sort_indices()
            cur.append(m)
        out.append(cur)
    return out


# ------------------------------------------------------------------------------ aggregation
def spmm(adj, x):
    """layers.py:43,45 torch.sparse.mm(adj, x) for one matrix."""
    indptr, indices, val = _canon(adj)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    y = np.empty((adj.shape[0], d), dtype=np.float32)
    rc = lib().oracle_spmm_csr_f32(ctypes.c_int64(adj.shape[0]), ctypes.c_int64(d), _p(indptr), _p(indices),
                                   _p(val), _p(x), _p(y), ctypes.c_int(0))
    assert rc == 0
    return y


def _ptr_arrays(adj_list):
    trip = [_canon(a) for a in adj_list]
    K = len(trip)
    mk = lambda i: (ctypes.c_void_p * K)(*[t[i].ctypes.data for t in trip])
    return trip, mk(0), mk(1), mk(2)


def core_aggregate(adj_list, x, return_pre=False):
    """layers.py:41-48 and :58 — cumulative SpMM over the k-core list, ReLU, laid out [N, K, d]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    K = len(adj_list)
    trip, a, b, c = _ptr_arrays(adj_list)
    H = np.empty((n, K, d), dtype=np.float32)
    pre = np.empty((n, K, d), dtype=np.float32)
    rc = lib().oracle_core_aggregate_f32(ctypes.c_int64(n), ctypes.c_int64(d), ctypes.c_int32(K), a, b, c,
                                         _p(x), _p(H), _p(pre))
    assert rc == 0
    return (H, pre) if return_pre else H


def core_aggregate_bwd(adj_list, x, dH):
    """d(sum(H*dH))/dx of core_aggregate — what autograd derives from layers.py:41-48."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    dH = np.ascontiguousarray(dH, dtype=np.float32)
    n, d = x.shape
    K = len(adj_list)
    _, pre = core_aggregate(adj_list, x, return_pre=True)
    trip, a, b, c = _ptr_arrays(adj_list)
    dX = np.empty((n, d), dtype=np.float32)
    rc = lib().oracle_core_aggregate_bwd_f32(ctypes.c_int64(n), ctypes.c_int64(d), ctypes.c_int32(K), a, b, c,
                                             _p(pre), _p(dH), _p(dX))
    assert rc == 0
    return dX


def aggregated_edges(adj_list):
    """BASELINE metric numerator for one CoreDiffusion call: sum_k nnz(A_k)."""
    return int(sum(sp.csr_matrix(a).nnz for a in adj_list))


# ----------------------------------------------------------------- random-walk corpus
def negative_table(freq):
    """preprocessing/random_walk.py:54-60, scalar loop as written there: node i repeated int(((f_i / tot)**0.75) / Z) times."""
    freq = np.asarray(freq, dtype=np.int64)
    tot = freq.sum()
    out = []
    for i in range(len(freq)):
        out += [i] * int(((freq[i] / tot) ** 0.75) / 0.00001)
    return np.array(out, dtype=np.int64)


def matching_walk_outputs(adj, walk_length, walk_time):
    """Exact outputs of preprocessing/random_walk.py:8-69 on a graph where every non-isolated node has exactly one
    neighbour (walks are then deterministic): (pair matrix CSR, node frequencies)."""
    adj = sp.csr_matrix(adj)
    n = adj.shape[0]
    freq = np.zeros(n, dtype=np.int64)
    rows, cols = [], []
    L1 = walk_length + 1
    for v in range(n):
        nb = adj.indices[adj.indptr[v]:adj.indptr[v + 1]]
        assert len(nb) <= 1
        if len(nb) == 0:
            continue
        walk = [v if i % 2 == 0 else int(nb[0]) for i in range(L1)]
        for _ in range(walk_time):
            for i in range(L1):
                for j in range(i + 1, L1):
                    if walk[i] != walk[j]:
                        rows += [walk[i], walk[j]]; cols += [walk[j], walk[i]]
                        freq[walk[i]] += 1; freq[walk[j]] += 1
    m = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    m.data[:] = 1.0
    m.sum_duplicates(); m.data[:] = 1.0
    m.sort_indices()
    return m, freq


def neg_sampling_loss(emb_list, node_idx_list, pos_idx_list, neg_idx_list, Q):
    """metrics.py:38-60 given the drawn indices: sum over snapshots of BCE(pos,1) + Q * BCE(neg,0), shape [1]."""
    import torch
    bce = torch.nn.BCEWithLogitsLoss()
    loss = torch.zeros(1)
    for emb, ni, pi, gi in zip(emb_list, node_idx_list, pos_idx_list, neg_idx_list):
        if ni is None or len(ni) == 0:
            continue
        pos = torch.sum(emb[ni].mul(emb[pi]), dim=1)
        neg = torch.sum(emb[ni].matmul(emb[gi].t()), dim=1)
        loss = loss + bce(pos, torch.ones_like(pos)) + Q * bce(neg, torch.zeros_like(neg))
    return loss
