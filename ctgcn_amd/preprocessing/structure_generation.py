"""k-core structure generation with the reference's class/constructor/method names
(reference preprocessing/structure_generation.py:11-81), computed by the HIP k-core peel.

For every snapshot file under <base>/<origin_folder> it writes <base>/<core_folder>/<snapshot>/<kk>.npz for
k = 1..max_core: the N x N scipy CSR of the subgraph induced by {v : core[v] >= k}, rows/columns in nodes-file
order, weights kept — the interchange format reference helper.py:69 reads.  One integer peel + one level tag
per edge replaces the reference's K networkx subgraph copies; the K files are then cut from the tagged CSR.

Optionally (cache_folder=...) one binary cache file per snapshot is written as well / instead:
<base>/<cache_folder>/<snapshot>.coreadj.npz = the snapshot's symmetric CSR (int32 indptr / indices, weights) + the int32
core numbers.  DataLoader.get_core_adj_list reads such a folder directly (no per-k files, no second peel): every k-core
matrix of every max_core setting is a level filter min(core[u], core[v]) >= k over that one CSR.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch

from .. import ops
from ..utils import check_and_make_path, get_format_str, read_edge_rows, symmetric_csr_from_rows


class StructureInfoGenerator(object):
    def __init__(self, base_path, origin_folder, core_folder, node_file):
        self.base_path = base_path
        self.origin_base_path = os.path.abspath(os.path.join(base_path, origin_folder))
        self.core_base_path = os.path.abspath(os.path.join(base_path, core_folder))
        with open(os.path.abspath(os.path.join(base_path, node_file)), 'r') as fp:
            self.full_node_list = [line.rstrip('\n') for line in fp if line.rstrip('\n') != '']
        self.node_num = len(self.full_node_list)
        self.node2idx = dict(zip(self.full_node_list, range(self.node_num)))
        check_and_make_path(self.core_base_path)

    def snapshot_csr(self, input_file, sep='\t'):
        src, dst, w = read_edge_rows(os.path.join(self.origin_base_path, input_file), self.node2idx, sep)
        return symmetric_csr_from_rows(src, dst, w, self.node_num)

    def core_numbers(self, csr, device='cuda'):
        """(core int32[N] numpy, max core, level int32[nnz] numpy) via the HIP peel."""
        row_ptr = torch.from_numpy(csr.indptr.astype(np.int32)).to(device)
        col = torch.from_numpy(csr.indices.astype(np.int32)).to(device)
        core, max_core = ops.kcore(row_ptr, col)
        if csr.nnz:
            val = torch.from_numpy(csr.data.astype(np.float32)).to(device)
            level, _, _ = ops.edge_levels(row_ptr, col, val, core, max_core + 1)
            level = level.cpu().numpy()
        else:
            level = np.zeros(0, dtype=np.int32)
        return core.cpu().numpy(), max_core, level

    def get_kcore_graph(self, input_file, output_dir, sep='\t', core_list=None, degree_list=None, cache_file=None,
                        per_k_files=True):
        csr = self.snapshot_csr(input_file, sep)
        core, max_core_num, level = self.core_numbers(csr)
        print("unique core nums: ", len(np.unique(core)))
        print('file name: ', input_file, 'max core num: ', max_core_num)
        if per_k_files:
            check_and_make_path(output_dir)
            fmt = get_format_str(max_core_num)
            rows = np.repeat(np.arange(self.node_num), np.diff(csr.indptr))
            for k in range(1, max_core_num + 1):
                keep = level >= k
                sub = sp.csr_matrix((csr.data[keep], (rows[keep], csr.indices[keep])), shape=csr.shape)
                sub.sort_indices()
                sp.save_npz(os.path.join(output_dir, fmt.format(k) + '.npz'), sub)
        if cache_file is not None:
            write_core_cache(cache_file, csr, core)
        if core_list is not None:
            core_list.append(max_core_num)
        return core

    def get_kcore_graph_all_time(self, sep='\t', worker=-1, cache_folder=None, per_k_files=True):
        """`worker` is accepted for signature compatibility; snapshots are processed in order on the one GPU
        (the reference forks a process pool because its per-snapshot work is pure Python).
        cache_folder: also write <base>/<cache_folder>/<snapshot>.coreadj.npz (one file per snapshot, see the module
        docstring); per_k_files=False then skips the reference's K files per snapshot."""
        print("getting k-core sub-graphs for all timestamps...")
        cache_base = None
        if cache_folder is not None:
            cache_base = os.path.abspath(os.path.join(self.base_path, cache_folder))
            check_and_make_path(cache_base)
        for f_name in sorted(os.listdir(self.origin_base_path)):
            stem = f_name.split('.')[0]
            self.get_kcore_graph(input_file=f_name, output_dir=os.path.join(self.core_base_path, stem), sep=sep,
                                 cache_file=None if cache_base is None else os.path.join(cache_base, stem + CACHE_SUFFIX),
                                 per_k_files=per_k_files)
        print("got it...")


CACHE_SUFFIX = '.coreadj.npz'


def write_core_cache(path, csr, core):
    """One snapshot = symmetric zero-diagonal CSR (rows/columns in nodes-file order, sorted indices) + core numbers."""
    csr = sp.csr_matrix(csr)
    csr.sort_indices()
    if csr.nnz >= 2 ** 31:
        raise ValueError("more than 2^31-1 stored entries")
    with open(path, 'wb') as fp:       # file object: numpy must not append another '.npz'
        np.savez(fp, indptr=csr.indptr.astype(np.int32), indices=csr.indices.astype(np.int32), data=csr.data,
                 core=np.asarray(core, dtype=np.int32), format=np.array([1], dtype=np.int32))


def read_core_cache(path):
    """-> (indptr int32, indices int32, data, core int32)"""
    with np.load(path, allow_pickle=False) as z:
        if int(z['format'][0]) != 1:
            raise ValueError("unknown core cache format in %s" % path)
        return z['indptr'], z['indices'], z['data'], z['core']
