"""Host-side helpers around the hot path: edge-list ingestion with the reference's graph semantics
(reference utils.py:23-58) and the file-name convention of the per-k .npz files (utils.py:142-148).
"""
import os

import numpy as np
import scipy.sparse as sp


def check_and_make_path(path):
    if path and not os.path.exists(path):
        os.makedirs(path)


def get_format_str(cnt):
    """'{:0>Wd}' with W = number of decimal digits of cnt, so sorted(file names) is numeric order."""
    return '{:0>' + str(len(str(int(cnt))) if cnt > 0 else 0) + 'd}'


def read_edge_rows(file_path, node2idx, sep='\t'):
    """Parse a snapshot file `from_id<sep>to_id[<sep>weight]` with a header line into index arrays
    (row order preserved — it decides which duplicate wins).  Unweighted files get weight 1."""
    src, dst, w = [], [], []
    with open(file_path, 'r') as fp:
        next(fp, None)
        for line in fp:
            parts = line.rstrip('\n').split(sep)
            if len(parts) < 2 or parts[0] == '':
                continue
            assert len(parts) in (2, 3)
            src.append(node2idx[parts[0]])
            dst.append(node2idx[parts[1]])
            w.append(float(parts[2]) if len(parts) == 3 else 1.0)
    return np.asarray(src, dtype=np.int64), np.asarray(dst, dtype=np.int64), np.asarray(w, dtype=np.float64)


def symmetric_csr_from_rows(src, dst, w, n):
    """Undirected simple weighted graph, vectorised: the LAST row naming an unordered pair {u, v} sets its
    weight, self loops are dropped (what nx.from_pandas_edgelist + remove self loops does at reference
    utils.py:23-30, and what the A[i,j]=A[j,i]=w overwrite loop does at utils.py:49-56).
    Returns scipy CSR float64, symmetric, zero diagonal, sorted indices."""
    src, dst, w = np.asarray(src, np.int64), np.asarray(dst, np.int64), np.asarray(w, np.float64)
    keep = src != dst
    src, dst, w = src[keep], dst[keep], w[keep]
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    key = lo * n + hi
    # stable sort by key, then the last element of each run is the winner
    order = np.argsort(key, kind='stable')
    key_s = key[order]
    last = np.ones(len(key_s), dtype=bool)
    last[:-1] = key_s[1:] != key_s[:-1]
    win = order[last]
    lo, hi, w = lo[win], hi[win], w[win]
    m = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([lo, hi]), np.concatenate([hi, lo]))), shape=(n, n)).tocsr()
    m.sort_indices()
    return m


def get_sp_adj_mat(file_path, full_node_list, sep='\t'):
    """Same result as the reference's get_sp_adj_mat (utils.py:35-58), returned as scipy COO."""
    node2idx = dict(zip(full_node_list, range(len(full_node_list))))
    src, dst, w = read_edge_rows(file_path, node2idx, sep)
    return symmetric_csr_from_rows(src, dst, w, len(full_node_list)).tocoo()
