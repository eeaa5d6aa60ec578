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


class NodeIndex(object):
    """Node name -> row index join for whole columns at once (Arrow hash join), built once per node list."""

    def __init__(self, node_list):
        import pyarrow as pa
        self.names = pa.array([str(x) for x in node_list], type=pa.string())
        self.n = len(node_list)

    def lookup(self, columns, file_path):
        """int64 index arrays of the given name columns.  index_in is a single-threaded hash probe: the columns are cut into a
        few slices and probed concurrently (pyarrow releases the GIL; every call builds its own table of the node names)."""
        import pyarrow.compute as pc
        from concurrent.futures import ThreadPoolExecutor
        parts = max(1, min((os.cpu_count() or 1) // max(1, len(columns)), 8))
        jobs = []
        for ci, col in enumerate(columns):
            step = -(-len(col) // parts) if len(col) else 1
            jobs += [(ci, lo, col.slice(lo, step)) for lo in range(0, max(len(col), 1), step)]

        def probe(job):
            ci, lo, piece = job
            idx = pc.index_in(piece, value_set=self.names)
            if idx.null_count:
                bad = piece.filter(pc.is_null(idx))[0].as_py()
                raise KeyError("%s: node %r is not in the node list" % (file_path, bad))
            return ci, lo, idx.to_numpy(zero_copy_only=False)

        out = [np.empty(len(col), dtype=np.int64) for col in columns]
        if len(jobs) == 1:
            results = [probe(jobs[0])]
        else:
            with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
                results = list(ex.map(probe, jobs))
        for ci, lo, arr in results:
            out[ci][lo:lo + len(arr)] = arr
        return out


_index_cache = {}


def _node_index(node2idx):
    """NodeIndex of a name -> index dict whose values are 0..n-1 in insertion order (how every caller builds it); cached by
    identity (the cache entry keeps the dict alive, so the id cannot be recycled)."""
    hit = _index_cache.get(id(node2idx))
    if hit is None or hit[0] is not node2idx:
        if len(_index_cache) >= 3:        # a run has one node list (DataLoader, StructureInfoGenerator and WalkGenerator each hold a dict of it)
            _index_cache.clear()
        vals = np.fromiter(node2idx.values(), dtype=np.int64, count=len(node2idx))
        if not np.array_equal(vals, np.arange(len(vals))):
            return None
        hit = _index_cache[id(node2idx)] = (node2idx, NodeIndex(list(node2idx.keys())))
    return hit[1]


def read_edge_rows(file_path, node2idx, sep='\t'):
    """Parse a snapshot file `from_id<sep>to_id[<sep>weight]` with a header line into index arrays
    (row order preserved — it decides which duplicate wins).  Unweighted files get weight 1.
    Columnar: Arrow's multi-threaded CSV reader + one hash join per endpoint column (an 8 M-row snapshot in about a second;
    the per-line Python loop this replaces took minutes).  Falls back to the line loop without pyarrow."""
    try:
        import pyarrow as pa
        import pyarrow.csv as pacsv
        index = _node_index(node2idx)
    except ImportError:
        index = None
    if index is None:
        return _read_edge_rows_loop(file_path, node2idx, sep)
    with open(file_path, 'r') as fp:
        header = fp.readline().rstrip('\n').rstrip('\r')
        first = fp.readline()
    if header == '' or first == '':
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float64)      # header only / empty file
    names = header.split(sep)
    assert len(names) in (2, 3)
    if len(set(names)) != len(names):
        return _read_edge_rows_loop(file_path, node2idx, sep)
    types = {names[0]: pa.string(), names[1]: pa.string()}
    if len(names) == 3:
        types[names[2]] = pa.float64()
    tbl = pacsv.read_csv(file_path, parse_options=pacsv.ParseOptions(delimiter=sep, quote_char=False),
                         convert_options=pacsv.ConvertOptions(column_types=types, strings_can_be_null=False))
    src_col, dst_col = tbl.column(0).combine_chunks(), tbl.column(1).combine_chunks()
    src, dst = index.lookup([src_col, dst_col], file_path)
    if len(names) == 3:
        w = tbl.column(2).to_numpy().astype(np.float64)
    else:
        w = np.ones(len(src), dtype=np.float64)
    return src, dst, w


def _read_edge_rows_loop(file_path, node2idx, sep='\t'):
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


_dict_cache = {}


def _node_dict(full_node_list):
    """name -> index dict of a node list, ONE per list object (callers pass the same list for every snapshot file; a fresh dict per call
    also missed _node_index's identity cache and rebuilt the Arrow name array per file).  The entry keeps the list alive (its id cannot
    be recycled) and is rebuilt when the list was mutated in place: a shallow copy taken at build time is compared element by element
    (identity shortcut per element, ~1 ms per million names), so reordered / renamed nodes never meet a stale mapping — the reference
    rebuilds the dict on every call (utils.py:37)."""
    hit = _dict_cache.get(id(full_node_list))
    if hit is None or hit[0] is not full_node_list or hit[2] != full_node_list:
        if len(_dict_cache) >= 2:
            _dict_cache.clear()
        hit = _dict_cache[id(full_node_list)] = (full_node_list, dict(zip(full_node_list, range(len(full_node_list)))), list(full_node_list))
    return hit[1]


def get_sp_adj_mat(file_path, full_node_list, sep='\t'):
    """Same result as the reference's get_sp_adj_mat (utils.py:35-58), returned as scipy COO."""
    src, dst, w = read_edge_rows(file_path, _node_dict(full_node_list), sep)
    return symmetric_csr_from_rows(src, dst, w, len(full_node_list)).tocoo()
