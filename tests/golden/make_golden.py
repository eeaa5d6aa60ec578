#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

This script is the only place in the repo that touches /root/reference, and it
only runs in the build container (the GPU box has no /root/reference).  It
imports the reference's Python modules in-process, feeds them inputs, and
writes INPUTS + EXPECTED OUTPUTS as .npz data.  No reference source text is
stored.  Re-run:  python tests/golden/make_golden.py

Run-time shims (reference files untouched):
  * np.int = int                          (numpy 2 removed the alias)
  * nx.to_scipy_sparse_matrix             (removed in networkx 3)
  * np.random.normal(loc=1x1 np.matrix)   (gen_degree_features only: helper.py:131 passes a 1x1 matrix as loc, which
                                           numpy rejects together with size=; the shim hands numpy float(loc))

Outputs
  uci_snapshots.npz      the bundled UCI snapshot edge lists as index arrays
  uci_kcore.npz          core numbers + every per-k .npz the reference writes
  uci_core_adj.npz       DataLoader.get_core_adj_list outputs, max_core in {-1, 5}
  weighted_small.npz     3 tiny weighted graphs w/ duplicates and self loops:
                         k-core files, loader outputs, CoreDiffusion fwd/bwd
  models_uci.npz         CGCN-C/S and CTGCN-C/S forward (+ input grads) on UCI
  toy_kcore.npz          hand-sized k-core known answers
  export_tsv.npz         bytes of the file the reference's save_embedding writes for a crafted embedding
  degree_features.npz    DataLoader.get_degree_feature_list (one-hot / adj on UCI; all four init types on a small weighted
                         graph under np.random.seed) and get_feature_list on feature files
  models_w128.npz        hidden = embed = 128 (the width of every shipped config and of the fused HIP GRU kernels):
                         CoreDiffusion(128,128) forward + all gradients, CTGCN-C(24,128,128,1,2,3) and
                         CTGCN-S(24,128,128,3,1,3) forward + gradient checksums on a UCI window
  negloss.npz            reference random_walk outputs (deterministic matching graph; seeded UCI statistics) and a
                         NegativeSamplingLoss value + gradients on a draw-independent configuration
"""
import os
import shutil
import sys
import tempfile
import warnings

import numpy as np
import scipy.sparse as sp
import networkx as nx
import torch

warnings.filterwarnings("ignore")
np.int = int  # shim 1
nx.to_scipy_sparse_matrix = lambda G, nodelist=None: sp.csr_matrix(  # shim 2
    nx.to_scipy_sparse_array(G, nodelist=nodelist))

REF = "/root/reference"
sys.path.insert(0, REF)
from preprocessing.structure_generation import StructureInfoGenerator  # noqa: E402
from helper import DataLoader  # noqa: E402
import layers as ref_layers  # noqa: E402
import models as ref_models  # noqa: E402
import utils as ref_utils  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def csr_of_tensor(t):
    """torch sparse COO (uncoalesced) -> canonical scipy CSR float32 (dups summed)."""
    t = t.coalesce()
    idx = t.indices().numpy()
    m = sp.csr_matrix((t.values().numpy().astype(np.float32), (idx[0], idx[1])), shape=tuple(t.shape))
    m.sort_indices()
    return m


def put_csr(d, key, m):
    m = sp.csr_matrix(m)
    m.sort_indices()
    d[key + "_indptr"] = m.indptr.astype(np.int32)
    d[key + "_indices"] = m.indices.astype(np.int32)
    d[key + "_data"] = m.data.astype(np.float64)


def run_structure(base, node_names, snapshot_rows):
    """Write snapshots as the reference expects, run its k-core generator, return paths."""
    os.makedirs(os.path.join(base, "1.format"))
    os.makedirs(os.path.join(base, "nodes_set"))
    with open(os.path.join(base, "nodes_set", "nodes.csv"), "w") as fp:
        fp.write("\n".join(node_names) + "\n")
    for fname, (src, dst, w) in snapshot_rows.items():
        with open(os.path.join(base, "1.format", fname), "w") as fp:
            fp.write("from_id\tto_id\tweight\n")
            for s, t, ww in zip(src, dst, w):
                fp.write("%s\t%s\t%s\n" % (node_names[s], node_names[t], repr(float(ww)) if ww != int(ww) else str(int(ww))))
    gen = StructureInfoGenerator(base, "1.format", "2.core", "nodes_set/nodes.csv")
    devnull = open(os.devnull, "w")
    old = sys.stdout
    sys.stdout = devnull
    try:
        gen.get_kcore_graph_all_time(sep="\t", worker=-1)
    finally:
        sys.stdout = old
    return gen


def dump_core_files(d, prefix, core_dir):
    """Store every <snapshot>/<kk>.npz the reference wrote."""
    snaps = sorted(os.listdir(core_dir))
    d[prefix + "snapshots"] = np.array(snaps)
    for ti, s in enumerate(snaps):
        files = sorted(os.listdir(os.path.join(core_dir, s)))
        d[prefix + "t%d_files" % ti] = np.array(files)
        for f in files:
            m = sp.load_npz(os.path.join(core_dir, s, f))
            assert sp.isspmatrix_csr(m)
            put_csr(d, prefix + "t%d_%s" % (ti, f[:-4]), m)


def dump_core_adj(d, prefix, adj_list):
    d[prefix + "K"] = np.array([len(a) for a in adj_list], dtype=np.int32)
    for ti, inner in enumerate(adj_list):
        for j, a in enumerate(inner):
            assert not a.is_coalesced()
            assert a.dtype == torch.float32 and a._indices().dtype == torch.int64
            put_csr(d, prefix + "t%d_j%d" % (ti, j), csr_of_tensor(a))


# --------------------------------------------------------------------------- UCI
def gen_uci():
    src_dir = os.path.join(REF, "data", "uci")
    names = [l.strip() for l in open(os.path.join(src_dir, "nodes_set", "nodes.csv")) if l.strip()]
    name2idx = {n: i for i, n in enumerate(names)}
    files = sorted(os.listdir(os.path.join(src_dir, "1.format")))
    snap = {"node_names": np.array(names), "files": np.array(files)}
    rows = {}
    for ti, f in enumerate(files):
        lines = open(os.path.join(src_dir, "1.format", f)).read().split("\n")[1:]
        lines = [l.split("\t") for l in lines if l]
        src = np.array([name2idx[l[0]] for l in lines], dtype=np.int32)
        dst = np.array([name2idx[l[1]] for l in lines], dtype=np.int32)
        w = np.array([float(l[2]) for l in lines], dtype=np.float64)
        snap["t%d_src" % ti], snap["t%d_dst" % ti], snap["t%d_w" % ti] = src, dst, w
        rows[f] = (src, dst, w)
    np.savez_compressed(os.path.join(OUT, "uci_snapshots.npz"), **snap)

    tmp = tempfile.mkdtemp(prefix="golden_uci_")
    try:
        # run the reference on ITS OWN files (symlinks), not on our re-serialisation
        base = os.path.join(tmp, "uci")
        os.makedirs(base)
        os.symlink(os.path.join(src_dir, "1.format"), os.path.join(base, "1.format"))
        os.symlink(os.path.join(src_dir, "nodes_set"), os.path.join(base, "nodes_set"))
        gen = StructureInfoGenerator(base, "1.format", "2.core", "nodes_set/nodes.csv")
        old = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            gen.get_kcore_graph_all_time(sep="\t", worker=-1)
        finally:
            sys.stdout = old
        kc = {}
        for ti, f in enumerate(files):
            g = ref_utils.get_nx_graph(os.path.join(base, "1.format", f), gen.full_node_list, sep="\t")
            cn = nx.core_number(g)
            kc["core_t%d" % ti] = np.array([cn[n] for n in gen.full_node_list], dtype=np.int32)
        dump_core_files(kc, "", os.path.join(base, "2.core"))
        np.savez_compressed(os.path.join(OUT, "uci_kcore.npz"), **kc)

        ca = {}
        dl = DataLoader(gen.full_node_list, len(files))
        for mc, tag in ((-1, "mcm1_"), (5, "mc5_")):
            dump_core_adj(ca, tag, dl.get_core_adj_list(os.path.join(base, "2.core"), 0, 7, max_core=mc))
        # a window that does not start at 0 (sticky max_core picks up snapshot 4's file count)
        dump_core_adj(ca, "w4_", dl.get_core_adj_list(os.path.join(base, "2.core"), 4, 3, max_core=-1))
        np.savez_compressed(os.path.join(OUT, "uci_core_adj.npz"), **ca)

        gen_models(base, gen.full_node_list, len(files))
    finally:
        shutil.rmtree(tmp)


# ------------------------------------------------------------- tiny weighted graphs
def gen_weighted_small():
    rng = np.random.default_rng(20260928)
    d = {}
    cases = [dict(n=24, m=90, din=8, dout=8, rnn="GRU", mc=-1),
             dict(n=64, m=420, din=128, dout=16, rnn="GRU", mc=3),
             dict(n=48, m=260, din=12, dout=20, rnn="LSTM", mc=-1)]
    d["n_cases"] = np.int32(len(cases))
    for ci, c in enumerate(cases):
        n, m = c["n"], c["m"]
        names = ["V%03d" % i for i in range(n)]
        # a dense-ish head so that several core levels exist; duplicates in both
        # orientations with DIFFERENT weights (last one wins); self loops; isolated tail
        live = n - 4
        p = 1.0 / (np.arange(live) + 3.0)
        p /= p.sum()
        src = rng.choice(live, size=m, p=p).astype(np.int32)
        dst = rng.choice(live, size=m, p=p).astype(np.int32)
        w = rng.integers(1, 6, size=m).astype(np.float64) * 0.5
        src[5], dst[5] = dst[2], src[2]
        w[5] = 7.5
        tmp = tempfile.mkdtemp(prefix="golden_small_")
        try:
            snaps = {"s0.csv": (src[: m // 2], dst[: m // 2], w[: m // 2]), "s1.csv": (src, dst, w)}
            gen = run_structure(tmp, names, snaps)
            p_ = "c%d_" % ci
            d[p_ + "n"] = np.int32(n)
            for si, (f, (a, b, ww)) in enumerate(sorted(snaps.items())):
                d[p_ + "s%d_src" % si], d[p_ + "s%d_dst" % si], d[p_ + "s%d_w" % si] = a, b, ww
                g = ref_utils.get_nx_graph(os.path.join(tmp, "1.format", f), names, sep="\t")
                cn = nx.core_number(g)
                d[p_ + "s%d_core" % si] = np.array([cn[x] for x in names], dtype=np.int32)
                # date adjacency the reference derives straight from the edge list (utils.get_sp_adj_mat)
                put_csr(d, p_ + "s%d_dateadj" % si, sp.csr_matrix(ref_utils.get_sp_adj_mat(
                    os.path.join(tmp, "1.format", f), names, sep="\t")))
            dump_core_files(d, p_ + "core_", os.path.join(tmp, "2.core"))
            dl = DataLoader(names, 2)
            adj = dl.get_core_adj_list(os.path.join(tmp, "2.core"), 0, 2, max_core=c["mc"])
            d[p_ + "max_core"] = np.int32(c["mc"])
            dump_core_adj(d, p_ + "adj_", adj)

            # CoreDiffusion forward / backward on snapshot 1
            torch.manual_seed(100 + ci)
            layer = ref_layers.CoreDiffusion(c["din"], c["dout"], rnn_type=c["rnn"])
            with torch.no_grad():
                layer.norm.weight.uniform_(0.5, 1.5)
                layer.norm.bias.uniform_(-0.5, 0.5)
            x = torch.randn(n, c["din"], requires_grad=True)
            gout = torch.randn(n, c["dout"])
            # raw aggregation stage exactly as the layer's loop produces it
            with torch.no_grad():
                res, hs = None, []
                for j, a in enumerate(adj[1]):
                    res = torch.sparse.mm(a, x) if j == 0 else res + torch.sparse.mm(a, x)
                    hs.append(torch.relu(res))
                d[p_ + "cd_agg"] = torch.stack(hs, 0).transpose(0, 1).contiguous().numpy()
            out = layer(x, adj[1])
            (out * gout).sum().backward()
            d[p_ + "cd_rnn"] = np.array(c["rnn"])
            d[p_ + "cd_x"], d[p_ + "cd_gout"] = x.detach().numpy(), gout.numpy()
            d[p_ + "cd_out"], d[p_ + "cd_dx"] = out.detach().numpy(), x.grad.numpy()
            for k, v in layer.state_dict().items():
                d[p_ + "cd_sd_" + k] = v.numpy()
            for k, v in layer.named_parameters():
                if v.grad is not None:
                    d[p_ + "cd_grad_" + k] = v.grad.numpy()
        finally:
            shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "weighted_small.npz"), **d)


# ----------------------------------------------------------------- models on UCI
def formula_tensor(shape, freq, phase):
    """Deterministic pseudo-data: float32(sin(freq * i + phase) + 0.5 * cos(0.013 * i)) over the flat index i."""
    i = np.arange(int(np.prod(shape)), dtype=np.float64)
    return (np.sin(freq * i + phase) + 0.5 * np.cos(0.013 * i)).astype(np.float32).reshape(shape)


def gen_models(base, node_list, tnum):
    d = {}
    n = len(node_list)
    dl = DataLoader(node_list, tnum)
    start, dur = 4, 3
    adj = dl.get_core_adj_list(os.path.join(base, "2.core"), start, dur, max_core=-1)
    x_onehot, in_dim = dl.get_feature_list(None, start, dur)
    assert in_dim == n
    # closed-form inputs / loss weights so the fixture stores only expected OUTPUTS (tests use the same formula)
    x_dense = list(torch.from_numpy(formula_tensor((dur, n, 24), 0.11, 0.3)))
    d["start"], d["duration"] = np.int32(start), np.int32(dur)

    def record(tag, model, xs, single=False):
        gsel = torch.from_numpy(formula_tensor((dur, n, model.output_dim), 0.37, 1.1))
        out = model(xs, adj) if not single else model(xs[0], adj[0])
        if model.model_type == "S":
            out, trans = out
            d[tag + "trans"] = torch.stack(list(trans)).detach().numpy() if not single else trans.detach().numpy()
        o = torch.stack(list(out)) if isinstance(out, (list, tuple)) else out
        d[tag + "out"] = o.detach().numpy()
        loss = (o * (gsel if not single else gsel[0])).sum()
        model.zero_grad()
        loss.backward()
        for k, v in model.state_dict().items():
            d[tag + "sd_" + k] = v.numpy()
        for k, v in model.named_parameters():
            d[tag + "grad_" + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()

    # only ONE model takes the N-wide one-hot input (its N x hid table dominates the fixture size)
    emb = 8
    torch.manual_seed(0)
    record("ctgcn_c_", ref_models.CTGCN(n, 16, emb, 1, 2, dur, rnn_type="GRU", model_type="C", trans_activate_type="L"), x_onehot)
    hid, emb = 16, 8
    torch.manual_seed(1)
    record("ctgcn_s_", ref_models.CTGCN(24, hid, emb, 3, 1, dur, rnn_type="GRU", model_type="S", trans_activate_type="N"), x_dense)
    torch.manual_seed(2)
    record("ctgcn_c_lstm_", ref_models.CTGCN(24, hid, emb, 1, 2, dur, rnn_type="LSTM", model_type="C", trans_activate_type="L"), x_dense)
    torch.manual_seed(3)
    record("cgcn_c_", ref_models.CGCN(24, hid, emb, 1, 2, rnn_type="GRU", model_type="C", trans_activate_type="L"), x_dense)
    torch.manual_seed(4)
    record("cgcn_s_", ref_models.CGCN(24, hid, emb, 3, 1, rnn_type="GRU", model_type="S", trans_activate_type="N"), x_dense)
    torch.manual_seed(5)
    record("cgcn_c_single_", ref_models.CGCN(24, hid, emb, 2, 3, rnn_type="GRU", model_type="C", trans_activate_type="N"), x_dense, single=True)
    np.savez_compressed(os.path.join(OUT, "models_uci.npz"), **d)


# ------------------------------------------------------------------ toy k-core
def gen_toy():
    d = {}
    toys = {
        # 4-clique + pendant path, plus a separate edge (shape of the reference's tests/conftest.py toy graph)
        "clique_path": (7, [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3), (3, 4), (5, 6)]),
        "empty": (5, []),
        "star": (6, [(0, i) for i in range(1, 6)]),
        "ring": (8, [(i, (i + 1) % 8) for i in range(8)]),
        "two_cliques_bridge": (9, [(a, b) for a in range(4) for b in range(a + 1, 4)]
                               + [(a, b) for a in range(4, 9) for b in range(a + 1, 9)] + [(3, 4)]),
        "petersen": (10, list(nx.petersen_graph().edges())),
    }
    rng = np.random.default_rng(5)
    for nn_, mm in ((200, 1500), (500, 1200), (1000, 12000)):
        e = rng.integers(0, nn_, size=(mm, 2))
        toys["rand_%d_%d" % (nn_, mm)] = (nn_, [tuple(x) for x in e])
    g = nx.barabasi_albert_graph(600, 5, seed=3)
    toys["ba_600_5"] = (600, list(g.edges()))
    d["names"] = np.array(sorted(toys))
    for name, (n, edges) in toys.items():
        g = nx.Graph()
        g.add_nodes_from(range(n))
        g.add_edges_from(edges)
        g.remove_edges_from(nx.selfloop_edges(g))
        cn = nx.core_number(g)
        d[name + "_n"] = np.int32(n)
        d[name + "_edges"] = np.array(edges, dtype=np.int32).reshape(-1, 2)
        d[name + "_core"] = np.array([cn[i] for i in range(n)], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "toy_kcore.npz"), **d)


# ------------------------------------------------------------------ embedding export
def gen_export():
    """Runs the reference trainer's save_embedding (embedding.py:79-89) on values that exercise every formatting branch
    and stores the bytes of the file it writes."""
    import embedding as ref_embedding
    rng = np.random.default_rng(11)
    n, dcols = 300, 9
    emb = rng.standard_normal((n, dcols)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -2.0, 0.1, 123456.789, 1e-5, 1.5e-5, 1e-4, 9.999e-5, 1.0001e-4, 1e15, 1e16, 1.234e20, 3.4e38,
                        1e-38, 1e-45, np.nan, np.inf, -np.inf, 16777216.0, 9999999.0, 99999.99, 0.30000001192092896, 5e-324,
                        2.5, 1e7, 1.17549435e-38, 65504.0, -7.0e-10], dtype=np.float32)
    emb.reshape(-1)[: len(special)] = special
    emb[50] = emb[50] * 1e-6
    emb[51] = emb[51] * 1e18
    emb[52] = np.round(emb[52] * 100)
    names = ["U%d" % i for i in range(n)]
    names[3], names[4], names[5] = "tab\there", 'quo"te', "plain name"
    tmp = tempfile.mkdtemp(prefix="golden_export_")
    try:
        obj = ref_embedding.BaseEmbedding.__new__(ref_embedding.BaseEmbedding)
        obj.timestamp_list = ["2004-04.csv", "2004-05.csv"]
        obj.full_node_list = names
        obj.embedding_base_path = tmp
        obj.file_sep = "\t"
        obj.save_embedding([torch.from_numpy(emb)], 1)
        data = open(os.path.join(tmp, "2004-05.csv"), "rb").read()
    finally:
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "export_tsv.npz"), emb=emb, names=np.array(names),
                        file_bytes=np.frombuffer(data, dtype=np.uint8))


# --------------------------------------------------- random-walk corpus + negative-sampling loss
def gen_negloss():
    import json
    import preprocessing.random_walk as ref_rw
    import metrics as ref_metrics
    d = {}
    # (1) a graph on which the reference's walks are deterministic: a perfect matching (every node has ONE neighbour)
    #     plus isolated nodes; weights arbitrary.  walk = a,b,a,b,...  -> pairs and frequencies are exact.
    n, L, W = 40, 5, 7
    a = np.arange(0, 30, 2)
    w = np.linspace(0.5, 3.0, len(a))
    adj = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([a, a + 1]), np.concatenate([a + 1, a]))), shape=(n, n))
    tmp = tempfile.mkdtemp(prefix="golden_walk_")
    try:
        os.makedirs(os.path.join(tmp, "pairs")); os.makedirs(os.path.join(tmp, "freq"))
        old = sys.stdout; sys.stdout = open(os.devnull, "w")
        try:
            ref_rw.random_walk(adj, os.path.join(tmp, "pairs"), os.path.join(tmp, "freq"), "m.csv", L, W, True)
        finally:
            sys.stdout = old
        m = sp.load_npz(os.path.join(tmp, "pairs", "m.npz")).tocsr(); m.sort_indices()
        put_csr(d, "match_adj", adj.tocsr()); put_csr(d, "match_pairs", m)
        d["match_neg"] = np.array(json.load(open(os.path.join(tmp, "freq", "m.json"))), dtype=np.int64)
        d["match_LW"] = np.array([L, W])
        # (2) distribution reference on a real snapshot (UCI 2004-10, the smallest): seeded numpy RNG, 10 walks per node
        names = [l.strip() for l in open(os.path.join(REF, "data", "uci", "nodes_set", "nodes.csv")) if l.strip()]
        uci = ref_utils.get_sp_adj_mat(os.path.join(REF, "data", "uci", "1.format", "2004-10.csv"), names, sep="\t")
        np.random.seed(12345)
        sys.stdout = open(os.devnull, "w")
        try:
            ref_rw.random_walk(uci, os.path.join(tmp, "pairs"), os.path.join(tmp, "freq"), "u.csv", 5, 10, True)
        finally:
            sys.stdout = old
        mu = sp.load_npz(os.path.join(tmp, "pairs", "u.npz")).tocsr()
        neg = np.array(json.load(open(os.path.join(tmp, "freq", "u.json"))), dtype=np.int64)
        d["uci_pairs_nnz"] = np.int64(mu.nnz)
        d["uci_pairs_rowcount"] = np.diff(mu.indptr).astype(np.int32)
        d["uci_neg_counts"] = np.bincount(neg, minlength=len(names)).astype(np.int32)
    finally:
        shutil.rmtree(tmp)
    # (3) the loss on a configuration whose draws are deterministic: every partner list <= neg_num (all are taken) and
    #     a negative table of exactly neg_num entries (random.sample returns a permutation; the score sums are invariant)
    rng = np.random.default_rng(3)
    n2, T, dim, neg_num, Q = 60, 2, 16, 5, 7
    pair_lists, tables = [], []
    for t in range(T):
        rows = np.empty(n2, dtype=object)
        for i in range(n2):
            k = int(rng.integers(0, neg_num + 1))
            rows[i] = sorted(rng.choice(n2, size=k, replace=False).tolist())
        pair_lists.append(rows)
        tables.append(rng.choice(n2, size=neg_num, replace=False).tolist())
    emb = [torch.from_numpy(rng.standard_normal((n2, dim)).astype(np.float32)).requires_grad_(True) for _ in range(T)]
    batch = torch.from_numpy(rng.choice(n2, size=25, replace=False).astype(np.int64))
    loss = ref_metrics.NegativeSamplingLoss(pair_lists, tables, neg_num=neg_num, Q=Q)([emb, batch])
    loss.backward()
    d["loss_value"] = loss.detach().numpy()
    d["loss_batch"] = batch.numpy()
    d["loss_cfg"] = np.array([n2, T, dim, neg_num, Q])
    for t in range(T):
        d["loss_emb%d" % t] = emb[t].detach().numpy()
        d["loss_grad%d" % t] = emb[t].grad.numpy()
        d["loss_table%d" % t] = np.array(tables[t], dtype=np.int64)
        d["loss_pairs%d_len" % t] = np.array([len(r) for r in pair_lists[t]], dtype=np.int32)
        d["loss_pairs%d_flat" % t] = np.array([c for r in pair_lists[t] for c in r], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "negloss.npz"), **d)



# ------------------------------------------------- degree features / feature files (CTGCN-S input path, helper.py:109-192)
def put_coo_tensor(d, key, t):
    """a torch sparse COO tensor exactly as the loader returned it (entry order kept, not coalesced)"""
    d[key + "_idx"] = t._indices().numpy().astype(np.int32)
    d[key + "_val"] = t._values().numpy()
    d[key + "_shape"] = np.array(t.shape, dtype=np.int64)
    assert t._values().dtype == torch.float32


def gen_degree_features():
    d = {}
    src_dir = os.path.join(REF, "data", "uci")
    names = [l.strip() for l in open(os.path.join(src_dir, "nodes_set", "nodes.csv")) if l.strip()]
    dl = DataLoader(names, 7)
    for it, tag in (("one-hot", "uci_onehot_"), ("adj", "uci_adj_")):
        xs, dim = dl.get_degree_feature_list(os.path.join(src_dir, "1.format"), 4, 3, init_type=it)
        d[tag + "dim"] = np.int64(dim)
        for t, x in enumerate(xs):
            put_coo_tensor(d, tag + "t%d" % t, x)
    # all four init types on weighted_small case 1 (n = 64; fractional weights -> astype(int) truncation), seeded numpy stream
    ws = np.load(os.path.join(OUT, "weighted_small.npz"))
    n = int(ws["c1_n"])
    wnames = ["V%03d" % i for i in range(n)]
    tmp = tempfile.mkdtemp(prefix="golden_deg_")
    orig_normal = np.random.normal
    try:
        os.makedirs(os.path.join(tmp, "1.format"))
        for si in range(2):
            with open(os.path.join(tmp, "1.format", "s%d.csv" % si), "w") as fp:
                fp.write("from_id\tto_id\tweight\n")
                for a, b, w in zip(ws["c1_s%d_src" % si], ws["c1_s%d_dst" % si], ws["c1_s%d_w" % si]):
                    fp.write("%s\t%s\t%s\n" % (wnames[a], wnames[b], repr(float(w)) if w != int(w) else str(int(w))))
        np.random.normal = lambda loc=0.0, scale=1.0, size=None: orig_normal(float(loc), scale, size)   # shim 3
        dl2 = DataLoader(wnames, 2)
        d["small_seed"], d["small_std"] = np.int64(77), np.float64(0.05)
        for it, tag in (("gaussian", "small_gaussian_"), ("combine", "small_combine_"), ("one-hot", "small_onehot_"), ("adj", "small_adj_")):
            np.random.seed(77)
            xs, dim = dl2.get_degree_feature_list(os.path.join(tmp, "1.format"), 0, 2, init_type=it, std=0.05)
            d[tag + "dim"] = np.int64(dim)
            for t, x in enumerate(xs):
                if x.is_sparse:
                    put_coo_tensor(d, tag + "t%d" % t, x)
                else:
                    assert x.dtype == torch.float32
                    d[tag + "t%d" % t] = x.numpy()
        # feature files of different widths (helper.py:173-191)
        os.makedirs(os.path.join(tmp, "feat"))
        for i, width in enumerate((3, 5)):
            arr = formula_tensor((n, width), 0.29 + i, 0.7)
            with open(os.path.join(tmp, "feat", "f%d.csv" % i), "w") as fp:
                fp.write("\t".join("c%d" % c for c in range(width)) + "\n")
                for row in arr:
                    fp.write("\t".join(repr(float(v)) for v in row) + "\n")
        xs, dim = dl2.get_feature_list(os.path.join(tmp, "feat"), 0, 2)
        d["feat_dim"] = np.int64(dim)
        for t, x in enumerate(xs):
            d["feat_t%d" % t] = x.numpy()
    finally:
        np.random.normal = orig_normal
        shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "degree_features.npz"), **d)


# --------------------------------------------- hidden = embed = 128: the width the fused HIP GRU kernels cover
def seeded_parameters(module, seed):
    """Overwrite every parameter with values from numpy's PCG64 stream (same helper in tests/conftest.py), so the fixture
    stores expected OUTPUTS only: uniform(-b, b), b = 1/sqrt(last dim) for matrices, 0.1 for vectors; LayerNorm weight
    around 1."""
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(module.named_parameters())):
            rng = np.random.default_rng(seed * 1000 + k)
            bound = 1.0 / np.sqrt(p.shape[-1]) if p.dim() > 1 else 0.1
            v = rng.uniform(-bound, bound, size=tuple(p.shape)).astype(np.float32)
            if name.endswith("norm.weight"):
                v = v * 5.0 + 1.0
            p.copy_(torch.from_numpy(v))


def gen_models_w128():
    """Reference outputs at the width of every shipped config (embed_dim 128).  Weights and inputs are closed-form /
    seeded-numpy (regenerated by the tests); stored: a row sample + checksums of the outputs, gradients in full when
    small, else as (sum, abs-sum, 256 sampled entries)."""
    d = {}
    ca = np.load(os.path.join(OUT, "uci_core_adj.npz"))
    n, dur = 1899, 3
    adj = []
    for t in range(dur):
        mats = []
        for j in range(int(ca["w4_K"][t])):
            m = sp.csr_matrix((ca["w4_t%d_j%d_data" % (t, j)].astype(np.float32), ca["w4_t%d_j%d_indices" % (t, j)],
                               ca["w4_t%d_j%d_indptr" % (t, j)]), shape=(n, n))
            mats.append(ref_utils.sparse_mx_to_torch_sparse_tensor(m))
        adj.append(mats)
    rows = np.arange(0, n, 7)                        # 272 sampled node rows
    d["rows"] = rows.astype(np.int32)

    def put_tensor(key, g, full_below=6000):
        g = g.detach().numpy()
        d[key + "__sum"] = np.float64(g.astype(np.float64).sum())
        d[key + "__abssum"] = np.float64(np.abs(g.astype(np.float64)).sum())
        if g.size <= full_below:
            d[key] = g
        else:
            flat = g.reshape(-1)
            pick = np.linspace(0, flat.size - 1, 256).astype(np.int64)
            d[key + "__pick"] = pick
            d[key + "__vals"] = flat[pick]

    # (1) one CoreDiffusion(128, 128) layer on snapshot 0 of the window (K = 4)
    layer = ref_layers.CoreDiffusion(128, 128, rnn_type="GRU")
    seeded_parameters(layer, 11)
    x = torch.from_numpy(formula_tensor((n, 128), 0.19, 0.2)).requires_grad_(True)
    gout = torch.from_numpy(formula_tensor((n, 128), 0.41, 0.9))
    out = layer(x, adj[0])
    (out * gout).sum().backward()
    d["cd_out_rows"] = out.detach().numpy()[rows]
    d["cd_dx_rows"] = x.grad.numpy()[rows]
    put_tensor("cd_out", out)
    put_tensor("cd_dx", x.grad)
    for k, v in layer.named_parameters():
        if v.grad is not None:
            put_tensor("cd_grad_" + k, v.grad)

    # (2) models
    x_dense = list(torch.from_numpy(formula_tensor((dur, n, 24), 0.11, 0.3)))

    def record(tag, model, seed, xs):
        seeded_parameters(model, seed)
        gsel = torch.from_numpy(formula_tensor((dur, n, model.output_dim), 0.37, 1.1))
        out = model(xs, adj)
        if model.model_type == "S":
            out, trans = out
            d[tag + "trans_rows"] = torch.stack(list(trans)).detach().numpy()[:, rows]
        d[tag + "out_rows"] = out.detach().numpy()[:, rows]
        put_tensor(tag + "out", out)
        model.zero_grad()
        (out * gsel).sum().backward()
        for k, v in model.named_parameters():
            put_tensor(tag + "grad_" + k, v.grad if v.grad is not None else torch.zeros_like(v))

    record("ctgcn_c_", ref_models.CTGCN(24, 128, 128, 1, 2, dur, rnn_type="GRU", model_type="C", trans_activate_type="L"), 21, x_dense)
    record("ctgcn_s_", ref_models.CTGCN(24, 128, 128, 3, 1, dur, rnn_type="GRU", model_type="S", trans_activate_type="N"), 22, x_dense)
    np.savez_compressed(os.path.join(OUT, "models_w128.npz"), **d)


if __name__ == "__main__":
    torch.set_num_threads(1)
    if len(sys.argv) > 1:                 # regenerate selected fixtures only:  make_golden.py degree w128
        for what in sys.argv[1:]:
            {"degree": gen_degree_features, "w128": gen_models_w128, "negloss": gen_negloss, "export": gen_export,
             "toy": gen_toy, "small": gen_weighted_small, "uci": gen_uci}[what]()
        sys.exit(0)
    gen_negloss()
    gen_export()
    gen_toy()
    gen_weighted_small()
    gen_uci()
    gen_degree_features()
    gen_models_w128()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print("%-24s %8.1f KiB" % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))
