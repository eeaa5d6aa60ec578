"""Seeded synthetic dynamic graphs for the benchmark configs (the real Enron/Facebook/AS/math files are not
available offline).  Follows SURVEY.md §8d: numpy default_rng(seed), a power-law (Chung-Lu style) endpoint
sampler with exponent 2.1, de-duplicated undirected edges without self loops, shuffled, and snapshot i = the
cumulative prefix of the shuffled edge list exactly as the reference's build_dynamic_graph cuts it
(reference graph.py:101-108: first `pos + base*i` rows).  Weights are 1.0.
"""
import numpy as np
import scipy.sparse as sp
import torch

DEFAULT_SEED = 20260928


def powerlaw_edges(n, n_edges, seed=DEFAULT_SEED, gamma=2.1, max_degree_hint=None):
    """Unique undirected edges (u < v), shuffled.  Endpoint i is drawn with probability ∝ (i + i0)^(-1/(gamma-1));
    i0 is chosen so that the largest expected degree is about max_degree_hint (real graphs of the
    reference's datasets have max degree 200..1500, README.md:168-176)."""
    rng = np.random.default_rng(seed)
    if max_degree_hint is None:
        max_degree_hint = min(2000, max(16, n // 20))
    alpha = 1.0 / (gamma - 1.0)
    ranks = np.arange(n, dtype=np.float64)
    lo, hi = 0.5, float(n)
    for _ in range(60):          # bisection on i0: expected max degree is monotone decreasing in i0
        i0 = 0.5 * (lo + hi)
        w = (ranks + i0) ** (-alpha)
        if 2.0 * n_edges * w[0] / w.sum() > max_degree_hint:
            lo = i0
        else:
            hi = i0
    w = (ranks + hi) ** (-alpha)
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(n)                      # hubs are not the low ids
    edges = np.zeros((0,), dtype=np.int64)
    need = n_edges
    while need > 0:
        m = int(need * 1.15) + 1024
        u = perm[np.minimum(np.searchsorted(cdf, rng.random(m)), n - 1)]
        v = perm[np.minimum(np.searchsorted(cdf, rng.random(m)), n - 1)]
        keep = u != v
        key = np.minimum(u, v)[keep].astype(np.int64) * n + np.maximum(u, v)[keep]
        edges = np.unique(np.concatenate([edges, key]))
        need = n_edges - len(edges)
    edges = rng.permutation(edges)[:n_edges]
    return (edges // n).astype(np.int64), (edges % n).astype(np.int64)


def prefix_sizes(total, snapshots):
    """Row counts of the cumulative snapshots (reference graph.py:96-108)."""
    base = total // snapshots
    first = base if total % snapshots == 0 else base + total % snapshots
    return [first + base * i for i in range(snapshots)]


def dynamic_graph(n, avg_deg, snapshots, seed=DEFAULT_SEED, max_degree_hint=None):
    """list[snapshots] of scipy CSR (symmetric, zero diagonal, weight 1.0, float64)."""
    u, v = powerlaw_edges(n, int(n * avg_deg / 2), seed, max_degree_hint=max_degree_hint)
    out = []
    for m in prefix_sizes(len(u), snapshots):
        uu, vv = u[:m], v[:m]
        a = sp.coo_matrix((np.ones(2 * m), (np.concatenate([uu, vv]), np.concatenate([vv, uu]))), shape=(n, n)).tocsr()
        a.sort_indices()
        out.append(a)
    return out


def snapshot_rows(n, n_edges, snapshots, seed=DEFAULT_SEED, cumulative=True, max_degree_hint=None):
    """Edge rows of every snapshot of a synthetic window as (u, v, [index array per snapshot]).

    cumulative=True   n_edges = size of the LAST snapshot; snapshot i = the first prefix_sizes()[i] rows of the shuffled
                      list (reference graph.py:101-108 — Enron / Facebook / math style growth).
    cumulative=False  n_edges = rows PER snapshot; every snapshot is an independent seeded draw of n_edges rows from a pool
                      of 2 x n_edges power-law edges (AS-style daily graphs: same node set, mostly-overlapping edge sets,
                      no growth; reference README.md:168-176)."""
    if cumulative:
        u, v = powerlaw_edges(n, n_edges, seed, max_degree_hint=max_degree_hint)
        return u, v, [np.arange(m) for m in prefix_sizes(len(u), snapshots)]
    u, v = powerlaw_edges(n, 2 * n_edges, seed, max_degree_hint=max_degree_hint)
    rng = np.random.default_rng(seed + 1)
    return u, v, [np.sort(rng.choice(len(u), size=n_edges, replace=False)) for _ in range(snapshots)]


def window_graph_device(n, n_edges, snapshots, device, seed=DEFAULT_SEED, cumulative=True, max_degree_hint=None, which=None):
    """{t: (row_ptr, col, val)} device CSR triples of the snapshots in `which` (default all) of snapshot_rows(...)."""
    u, v, picks = snapshot_rows(n, n_edges, snapshots, seed, cumulative, max_degree_hint)
    from . import ops
    ud, vd = torch.from_numpy(u.astype(np.int32)).to(device), torch.from_numpy(v.astype(np.int32)).to(device)
    out = {}
    for t in (range(snapshots) if which is None else which):
        if cumulative:
            m = len(picks[t])
            out[t] = ops.edges_to_csr(ud[:m], vd[:m], None, n)
        else:
            idx = torch.from_numpy(picks[t]).to(device)
            out[t] = ops.edges_to_csr(ud[idx], vd[idx], None, n)
    return out


def window_graph(n, n_edges, snapshots, seed=DEFAULT_SEED, cumulative=True, max_degree_hint=None):
    """Same windows as window_graph_device, as scipy CSR on the host (tests, CPU oracle)."""
    u, v, picks = snapshot_rows(n, n_edges, snapshots, seed, cumulative, max_degree_hint)
    out = []
    for idx in picks:
        uu, vv = u[idx], v[idx]
        a = sp.coo_matrix((np.ones(2 * len(uu)), (np.concatenate([uu, vv]), np.concatenate([vv, uu]))), shape=(n, n)).tocsr()
        a.sort_indices()
        out.append(a)
    return out


def dynamic_graph_device(n, avg_deg, snapshots, device, seed=DEFAULT_SEED, max_degree_hint=None, which=None):
    """Same graphs as dynamic_graph, built as device CSR triples (row_ptr int32, col int32, val float32) by the
    library's GPU ingest (ctgcn_edges_to_csr).  `which`: iterable of snapshot indices to build (default all);
    returns {t: (row_ptr, col, val)}."""
    u, v = powerlaw_edges(n, int(n * avg_deg / 2), seed, max_degree_hint=max_degree_hint)
    sizes = prefix_sizes(len(u), snapshots)
    from . import ops
    ud, vd = torch.from_numpy(u.astype(np.int32)).to(device), torch.from_numpy(v.astype(np.int32)).to(device)
    out = {}
    for t in (range(snapshots) if which is None else which):
        m = sizes[t]
        out[t] = ops.edges_to_csr(ud[:m], vd[:m], None, n)      # HIP ingest: symmetrise + sort + CSR
    return out
