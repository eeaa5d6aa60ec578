"""Embedding export with the reference's file contract (reference embedding.py:79-89, consumed by evaluation/*.py,
e.g. link_prediction.py:126-143): one `<timestamp>.csv` per snapshot, index = node names, header = 0..d-1, sep = '\\t',
written by the library's multi-threaded host formatter instead of pandas (byte-identical output)."""
import os

import numpy as np
import torch

from . import _lib


def _pack_names(node_list):
    enc = [str(n).encode("utf-8") for n in node_list]
    offsets = np.zeros(len(enc), dtype=np.int64)
    pos = 0
    for i, b in enumerate(enc):
        offsets[i] = pos
        pos += len(b) + 1
    return b"\0".join(enc) + b"\0", offsets


def write_embedding(path, embedding, node_list, sep='\t', threads=0):
    """embedding: [N, d] float32 tensor (any device) or numpy array."""
    if isinstance(embedding, torch.Tensor):
        embedding = embedding.detach().to("cpu", torch.float32).contiguous().numpy()
    emb = np.ascontiguousarray(embedding, dtype=np.float32)
    if emb.ndim != 2 or emb.shape[0] != len(node_list):
        raise ValueError("embedding must be [len(node_list), d]")
    blob, offsets = _pack_names(node_list)
    lib = _lib.load()
    rc = lib.ctgcn_write_embedding_tsv(os.fsencode(path), emb.shape[0], emb.shape[1], emb.ctypes.data, emb.shape[1], blob,
                                       offsets.ctypes.data, sep.encode("ascii"), threads)
    _lib.check(rc, "ctgcn_write_embedding_tsv")


def save_embedding(output_list, timestamp_list, start_idx, embedding_base_path, node_list, sep='\t'):
    """Same semantics as the reference trainer's save_embedding: a 2-D tensor is one static embedding, otherwise one file
    per leading index; file name = timestamp_list[start_idx + i] without its extension + '.csv'."""
    if isinstance(output_list, torch.Tensor) and output_list.dim() == 2:
        output_list = [output_list]
    os.makedirs(embedding_base_path, exist_ok=True)
    for i in range(len(output_list)):
        stamp = timestamp_list[start_idx + i].split('.')[0]
        write_embedding(os.path.join(embedding_base_path, stamp + '.csv'), output_list[i], node_list, sep=sep)
