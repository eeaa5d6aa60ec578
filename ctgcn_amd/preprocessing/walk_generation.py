"""WalkGenerator with the reference's constructor / method names (reference preprocessing/walk_generation.py:10-61),
producing the same two artefacts per snapshot file — `<walk_pair_folder>/<snapshot>.npz` (scipy COO, symmetric 0/1
co-occurrence matrix) and `<node_freq_folder>/<snapshot>.json` (the negative table) — from GPU random walks."""
import json
import os

import numpy as np
import scipy.sparse as sp
import torch

from .. import ops
from ..utils import check_and_make_path, read_edge_rows
from ..walks import negative_table, random_walk_corpus


class WalkGenerator(object):
    def __init__(self, base_path, origin_folder, walk_pair_folder, node_freq_folder, node_file, walk_time=100, walk_length=5, seed=None):
        """seed=None (the reference's behaviour: numpy's unseeded global RNG, random_walk.py:8-69): every snapshot and every run draws a
        fresh walk stream from OS entropy.  seed=int: reproducible, snapshot i of a run uses seed + i."""
        self.base_path = base_path
        self.origin_base_path = os.path.abspath(os.path.join(base_path, origin_folder))
        self.walk_pair_base_path = os.path.abspath(os.path.join(base_path, walk_pair_folder))
        self.node_freq_base_path = os.path.abspath(os.path.join(base_path, node_freq_folder))
        with open(os.path.abspath(os.path.join(base_path, node_file)), 'r') as fp:
            self.full_node_list = [line.rstrip('\n') for line in fp if line.rstrip('\n') != '']
        self.node2idx = dict(zip(self.full_node_list, range(len(self.full_node_list))))
        self.walk_time = walk_time
        self.walk_length = walk_length
        self.seed = seed
        self._calls = 0
        check_and_make_path(self.walk_pair_base_path)
        check_and_make_path(self.node_freq_base_path)

    def get_walk_info(self, f_name, original_graph_path, sep='\t', weighted=True, device='cuda'):
        n = len(self.full_node_list)
        src, dst, w = read_edge_rows(original_graph_path, self.node2idx, sep)
        dev = torch.device(device)
        row_ptr, col, val = ops.edges_to_csr(torch.from_numpy(src.astype(np.int32)).to(dev), torch.from_numpy(dst.astype(np.int32)).to(dev),
                                             torch.from_numpy(w.astype(np.float32)).to(dev), n)
        seed = int.from_bytes(os.urandom(8), "little") >> 1 if self.seed is None else int(self.seed) + self._calls
        self._calls += 1
        pairs, freq = random_walk_corpus(row_ptr, col, val, self.walk_length, self.walk_time, weighted=weighted, seed=seed)
        stem = f_name.split('.')[0]
        with open(os.path.join(self.node_freq_base_path, stem + '.json'), 'w') as fp:
            json.dump(negative_table(freq).tolist(), fp)
        sp.save_npz(os.path.join(self.walk_pair_base_path, stem + '.npz'), pairs.to_scipy().tocoo())
        return pairs, freq

    def get_walk_info_all_time(self, worker=-1, sep='\t', weighted=True):
        """`worker` is accepted for signature compatibility (the walks of one snapshot already fill the GPU)."""
        print("perform random walk for all file(s)...")
        for f_name in sorted(os.listdir(self.origin_base_path)):
            self.get_walk_info(f_name, os.path.join(self.origin_base_path, f_name), sep=sep, weighted=weighted)
