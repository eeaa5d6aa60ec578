"""CPU-only checks of the product's host logic (no kernel launches): ABI surface, loader semantics,
checkpoint schema, loud failure without a GPU, and that the product never touches the oracle."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import ROOT, load_golden, csr_from

import ctgcn_amd
from ctgcn_amd import CoreAdj, _lib
from ctgcn_amd.core_adj import slot_table


# ------------------------------------------------------------------------------------ C ABI surface
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ctgcn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ctgcn_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    import ctypes
    declared = _declared_symbols()
    assert len(declared) >= 10
    assert sorted(_lib.SIGNATURES) == declared, "binding and header disagree"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().ctgcn_abi_version() == _lib.ABI_VERSION
    assert _lib.load().ctgcn_workspace_bytes(_lib.OP_KCORE, 1000, 0, 0, 0) >= 1000 * 12


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "ctgcn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle|oracle/|libctgcn_oracle", src, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_cpu_tensors_fail_loudly():
    model = ctgcn_amd.CGCN(6, 8, 4, 1, 1)
    adj = CoreAdj.from_matrices([sp.eye(5, format="csr") + sp.csr_matrix(np.ones((5, 5)) - np.eye(5))])
    with pytest.raises(_lib.CtgcnHipError, match="no CPU fallback"):
        model(torch.randn(5, 6), adj)


# --------------------------------------------------------------------------------- CoreAdj, host side
def _golden_lists(tag):
    ca = load_golden("uci_core_adj.npz")
    n = 1899
    return [[csr_from(ca, tag + "t%d_j%d" % (t, j), n, np.float32) for j in range(int(k))] for t, k in enumerate(ca[tag + "K"])]


def _same(a, b):
    a, b = sp.csr_matrix(a), sp.csr_matrix(b)
    a.sort_indices(); b.sort_indices()
    assert a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.data.astype(np.float32), b.data.astype(np.float32))


def test_core_adj_round_trips_reference_loader_output():
    for mats in _golden_lists("mcm1_"):
        adj = CoreAdj.from_matrices(mats)
        assert adj.nested and adj.self_loop and adj.symmetric and adj.K == len(mats)
        assert adj.nnz_per_slot == [m.nnz for m in mats]
        assert adj.nnz == mats[-1].nnz - (adj.n if len(mats) == 1 else 0)    # one pass over the largest matrix
        for got, want in zip(adj.to_scipy_list(), mats):
            _same(got, want)
        # rows sorted by (slot, col)
        rp, sl, col = adj.row_ptr.numpy(), adj.slot.numpy().astype(np.int64), adj.col.numpy().astype(np.int64)
        rows = np.repeat(np.arange(adj.n), np.diff(rp))
        key = (rows * 256 + sl) * adj.n + col
        assert np.all(np.diff(key) > 0)


def test_core_adj_general_lists():
    rng = np.random.default_rng(0)
    n = 40
    mats = [sp.random(n, n, 0.1, random_state=i, format="csr", dtype=np.float32) for i in range(3)]
    adj = CoreAdj.from_matrices(mats)
    assert not adj.nested and not adj.self_loop
    for got, want in zip(adj.to_scipy_list(), mats):
        _same(got, want)
    # nested structure but a differing weight -> must NOT be fused
    a = mats[0]
    b = (a + mats[1]).tocsr()
    b2 = b.copy(); b2.data = b2.data * 1.5
    adj = CoreAdj.from_matrices([a, b2])
    assert not adj.nested
    for got, want in zip(adj.to_scipy_list(), [a, b2]):
        _same(got, want)
    with pytest.raises(ValueError):
        CoreAdj.from_matrices([])
    # torch sparse COO input, as the reference loader hands over (uncoalesced)
    coo = mats[0].tocoo()
    t = torch.sparse_coo_tensor(np.vstack((coo.row, coo.col)), coo.data, coo.shape)
    _same(CoreAdj.from_matrices([t]).to_scipy_list()[0], mats[0])


def test_data_loader_reads_reference_npz_layout(tmp_path):
    """write the reference's own per-k files (golden) to disk, read them with our DataLoader."""
    kc, ca = load_golden("uci_kcore.npz"), load_golden("uci_core_adj.npz")
    n = 1899
    base = tmp_path / "2.core"
    for t, snap in enumerate(kc["snapshots"]):
        os.makedirs(base / str(snap))
        for f in kc["t%d_files" % t]:
            sp.save_npz(str(base / str(snap) / str(f)), csr_from(kc, "t%d_%s" % (t, str(f)[:-4]), n).astype(np.int64))
    names = [str(x) for x in load_golden("uci_snapshots.npz")["node_names"]]
    dl = ctgcn_amd.DataLoader(names, 7)
    for tag, start, dur, mc in (("mcm1_", 0, 7, -1), ("mc5_", 0, 7, 5), ("w4_", 4, 3, -1)):
        got = dl.get_core_adj_list(str(base), start, dur, max_core=mc)
        assert [len(g) for g in got] == ca[tag + "K"].tolist()
        for t, adj in enumerate(got):
            for j, m in enumerate(adj.to_scipy_list()):
                _same(m, csr_from(ca, tag + "t%d_j%d" % (t, j), n, np.float32))
    with pytest.raises(AssertionError):
        dl.get_core_adj_list(str(base), 7, 1)


def test_slot_table_matches_loader_rules():
    """the level->slot table of the device route reproduces the file route on every UCI snapshot."""
    kc = load_golden("uci_kcore.npz")
    ca = load_golden("uci_core_adj.npz")
    n = 1899
    sticky = {"mcm1_": -1, "mc5_": 5}
    for tag, mc in sticky.items():
        cur = mc
        for t in range(7):
            core = kc["core_t%d" % t]
            a1 = csr_from(kc, "t%d_%s" % (t, str(kc["t%d_files" % t][0])[:-4]), n).tocoo()
            level = np.minimum(core[a1.row], core[a1.col])
            files = int(core.max())
            count = np.bincount(level, minlength=files + 1)
            wsum = np.bincount(level, weights=a1.data, minlength=files + 1)
            table, K, levels, nnz = slot_table(count, wsum, files, cur, n)
            if cur == -1:
                cur = files
            assert K == int(ca[tag + "K"][t])
            want = [csr_from(ca, tag + "t%d_j%d" % (t, j), n, np.float32) for j in range(K)]
            assert nnz == [m.nnz for m in want]
            ref = CoreAdj.from_matrices(want)
            slot = table[level]
            order = np.lexsort((a1.col, slot, a1.row))
            assert np.array_equal(ref.col.numpy(), a1.col[order]) and np.array_equal(ref.slot.numpy(), slot[order])
    with pytest.raises(NotImplementedError):
        slot_table([0, 2, 2], [0.0, 0.0, 2.0], 2, -1, 4)      # level-1 entries cancel to zero weight


def test_symmetric_csr_last_duplicate_wins():
    from ctgcn_amd.utils import symmetric_csr_from_rows
    g = load_golden("weighted_small.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        n = int(g[p + "n"])
        for s in range(2):
            got = symmetric_csr_from_rows(g[p + "s%d_src" % s], g[p + "s%d_dst" % s], g[p + "s%d_w" % s], n)
            _same(got, csr_from(g, p + "s%d_dateadj" % s, n))
    m = symmetric_csr_from_rows([0, 1, 2, 2], [1, 0, 2, 0], [1.0, 5.0, 9.0, 2.0], 3)
    assert m[0, 1] == 5.0 and m[1, 0] == 5.0 and m[2, 2] == 0 and m[0, 2] == 2.0


# ------------------------------------------------------------------------------- checkpoint schema
@pytest.mark.parametrize("tag,ctor", [
    ("ctgcn_c_", lambda: ctgcn_amd.CTGCN(1899, 16, 8, 1, 2, 3, rnn_type="GRU", model_type="C", trans_activate_type="L")),
    ("ctgcn_s_", lambda: ctgcn_amd.CTGCN(24, 16, 8, 3, 1, 3, rnn_type="GRU", model_type="S", trans_activate_type="N")),
    ("ctgcn_c_lstm_", lambda: ctgcn_amd.CTGCN(24, 16, 8, 1, 2, 3, rnn_type="LSTM", model_type="C", trans_activate_type="L")),
    ("cgcn_c_", lambda: ctgcn_amd.CGCN(24, 16, 8, 1, 2, rnn_type="GRU", model_type="C", trans_activate_type="L")),
    ("cgcn_s_", lambda: ctgcn_amd.CGCN(24, 16, 8, 3, 1, rnn_type="GRU", model_type="S", trans_activate_type="N")),
    ("cgcn_c_single_", lambda: ctgcn_amd.CGCN(24, 16, 8, 2, 3, rnn_type="GRU", model_type="C", trans_activate_type="N")),
])
def test_state_dict_schema_matches_reference_checkpoints(tag, ctor):
    g = load_golden("models_uci.npz")
    ref = {k[len(tag + "sd_"):]: g[k].shape for k in g.files if k.startswith(tag + "sd_")}
    model = ctor()
    mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert mine == ref
    model.load_state_dict({k: torch.from_numpy(g[tag + "sd_" + k]) for k in ref}, strict=True)
    assert model.method_name == {"ctgcn_c_": "CTGCN-C", "ctgcn_s_": "CTGCN-S", "ctgcn_c_lstm_": "CTGCN-C", "cgcn_c_": "CGCN-C",
                                 "cgcn_s_": "CGCN-S", "cgcn_c_single_": "CGCN-C"}[tag]


def test_constructor_error_conventions():
    with pytest.raises(ValueError, match="number of layers should be positive!"):
        ctgcn_amd.CDN(4, 4, 4, 0)
    with pytest.raises(AssertionError):
        ctgcn_amd.CoreDiffusion(4, 4, rnn_type="RNN")
    with pytest.raises(AssertionError):
        ctgcn_amd.CTGCN(4, 4, 4, 1, 1, 2, model_type="X")
    with pytest.raises(AssertionError):
        ctgcn_amd.MLP(4, 4, 4, 0)


def test_synthetic_generator_is_seeded_and_cumulative():
    from ctgcn_amd.synth import dynamic_graph, prefix_sizes
    a = dynamic_graph(3000, 8, 4, seed=3)
    b = dynamic_graph(3000, 8, 4, seed=3)
    assert [x.nnz for x in a] == [2 * s for s in prefix_sizes(12000, 4)]
    for x, y in zip(a, b):
        assert (x != y).nnz == 0
    for small, big in zip(a[:-1], a[1:]):          # snapshot i is a prefix of snapshot i+1 (graph.py:101-108)
        assert (small.multiply(big) != small).nnz == 0
    assert (a[-1] != a[-1].T).nnz == 0 and a[-1].diagonal().sum() == 0


# ------------------------------------------------------------------------------------- embedding export
def test_embedding_export_is_byte_identical_to_reference(tmp_path):
    """ctgcn_write_embedding_tsv (host C++, std::to_chars) vs the bytes the reference's save_embedding (pandas) wrote."""
    from ctgcn_amd import export
    g = load_golden("export_tsv.npz")
    names = [str(x) for x in g["names"]]
    want = bytes(g["file_bytes"])
    for threads in (1, 3, 0):
        path = tmp_path / ("out%d.csv" % threads)
        export.write_embedding(str(path), g["emb"], names, sep="\t", threads=threads)
        got = path.read_bytes()
        assert got == want, (threads, len(got), len(want))
    # the trainer-shaped entry point: one file per snapshot, named after the timestamp
    export.save_embedding(torch.from_numpy(g["emb"])[None], ["a.csv", "2004-05.csv"], 1, str(tmp_path / "emb"), names)
    assert (tmp_path / "emb" / "2004-05.csv").read_bytes() == want


def test_embedding_export_float_formatting_sweep(tmp_path):
    """numpy prints a float32 the same way pandas writes it: sweep magnitudes, signs and random mantissas."""
    from ctgcn_amd import export
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32).copy()
    vals = vals[np.isfinite(vals)][:16384].reshape(-1, 8)
    names = ["n%d" % i for i in range(len(vals))]
    path = tmp_path / "sweep.tsv"
    export.write_embedding(str(path), vals, names)
    lines = path.read_text().split("\n")[1:-1]
    assert len(lines) == len(vals)
    for row, line in zip(vals[::37], lines[::37]):
        assert line.split("\t")[1:] == [str(v) for v in row]


def test_negative_table_matches_reference_formula():
    """vectorised negative_table == the reference's scalar loop (oracle), incl. on the golden deterministic case."""
    from ctgcn_amd.walks import negative_table
    from oracle import oracle as O
    g = load_golden("negloss.npz")
    n = len(g["match_adj_indptr"]) - 1
    _, freq = O.matching_walk_outputs(csr_from(g, "match_adj", n), *[int(x) for x in g["match_LW"]])
    assert np.array_equal(negative_table(freq), g["match_neg"])
    rng = np.random.default_rng(0)
    for _ in range(5):
        f = rng.integers(0, 5000, 400) * (rng.random(400) < 0.8)
        assert np.array_equal(negative_table(f), O.negative_table(f))
    assert len(negative_table(np.zeros(7, dtype=np.int64))) == 0


# ------------------------------------------- CTGCN-S input path: degree features / feature files (helper.py:109-192)
def _coo_equal(t, g, key):
    assert t.is_sparse and t.dtype == torch.float32 and tuple(t.shape) == tuple(g[key + "_shape"])
    assert np.array_equal(t._indices().numpy().astype(np.int32), g[key + "_idx"])
    assert np.array_equal(t._values().numpy(), g[key + "_val"])


def _write_snapshot_files(folder, names, rows):
    os.makedirs(folder)
    for fname, (src, dst, w) in rows.items():
        with open(os.path.join(folder, fname), "w") as fp:
            fp.write("from_id\tto_id\tweight\n")
            for a, b, ww in zip(src, dst, w):
                fp.write("%s\t%s\t%s\n" % (names[a], names[b], repr(float(ww)) if ww != int(ww) else str(int(ww))))


def test_degree_feature_list_matches_reference_loader(tmp_path):
    """one-hot / adj on the UCI window and all four init types on a weighted graph (fractional weighted degrees, seeded
    numpy stream): same tensors, entry order, dtypes and input_dim as the reference's get_degree_feature_list."""
    g = load_golden("degree_features.npz")
    snaps = load_golden("uci_snapshots.npz")
    names = [str(x) for x in snaps["node_names"]]
    _write_snapshot_files(str(tmp_path / "uci"), names,
                          {str(f): (snaps["t%d_src" % t], snaps["t%d_dst" % t], snaps["t%d_w" % t]) for t, f in enumerate(snaps["files"])})
    dl = ctgcn_amd.DataLoader(names, 7)
    for it, tag in (("one-hot", "uci_onehot_"), ("adj", "uci_adj_")):
        xs, dim = dl.get_degree_feature_list(str(tmp_path / "uci"), 4, 3, init_type=it)
        assert dim == int(g[tag + "dim"]) and len(xs) == 3
        for t, x in enumerate(xs):
            _coo_equal(x, g, tag + "t%d" % t)
    ws = load_golden("weighted_small.npz")
    n = int(ws["c1_n"])
    wnames = ["V%03d" % i for i in range(n)]
    _write_snapshot_files(str(tmp_path / "small"), wnames,
                          {"s%d.csv" % si: (ws["c1_s%d_src" % si], ws["c1_s%d_dst" % si], ws["c1_s%d_w" % si]) for si in range(2)})
    dl2 = ctgcn_amd.DataLoader(wnames, 2)
    for it, tag in (("gaussian", "small_gaussian_"), ("combine", "small_combine_"), ("one-hot", "small_onehot_"), ("adj", "small_adj_")):
        np.random.seed(int(g["small_seed"]))
        xs, dim = dl2.get_degree_feature_list(str(tmp_path / "small"), 0, 2, init_type=it, std=float(g["small_std"]))
        assert dim == int(g[tag + "dim"])
        for t, x in enumerate(xs):
            if it == "gaussian":
                assert not x.is_sparse and x.dtype == torch.float32
                assert np.array_equal(x.numpy(), g[tag + "t%d" % t])       # same numpy stream, bit for bit
            else:
                _coo_equal(x, g, tag + "t%d" % t)
    with pytest.raises(AssertionError):
        dl2.get_degree_feature_list(str(tmp_path / "small"), 0, 2, init_type="degree")


def test_feature_file_loader_matches_reference(tmp_path):
    g = load_golden("degree_features.npz")
    from conftest import formula_tensor
    n = 64
    os.makedirs(tmp_path / "feat")
    for i, width in enumerate((3, 5)):
        arr = formula_tensor((n, width), 0.29 + i, 0.7)
        with open(tmp_path / "feat" / ("f%d.csv" % i), "w") as fp:
            fp.write("\t".join("c%d" % c for c in range(width)) + "\n")
            for row in arr:
                fp.write("\t".join(repr(float(v)) for v in row) + "\n")
    dl = ctgcn_amd.DataLoader(["V%03d" % i for i in range(n)], 2)
    xs, dim = dl.get_feature_list(str(tmp_path / "feat"), 0, 2)
    assert dim == int(g["feat_dim"]) == 5
    for t, x in enumerate(xs):
        assert x.dtype == torch.float32 and np.array_equal(x.numpy(), g["feat_t%d" % t])
    eye, dim = dl.get_feature_list(None, 0, 2)
    assert dim == n and eye[0].is_sparse and eye[0]._nnz() == n


# ------------------------------------------------------------------ forward-path selection (advisor finding, round 1)
def test_frozen_weights_with_input_gradients_take_the_autograd_path(monkeypatch):
    """All parameters frozen, inputs require grad, grad mode on: the write-in-place inference path must NOT be taken (it
    would feed an uninitialised buffer to the temporal head and drop the input gradients).  The HIP aggregation cannot
    run here; the CPU oracle is injected at the product's dispatch point as the checker."""
    from oracle import torch_path as TP, oracle as O
    from ctgcn_amd import ops
    from ctgcn_amd.synth import dynamic_graph

    def oracle_aggregate(x, adj, relu=True):
        mats = [TP.coo_like_reference(m) for m in adj.to_scipy_list()]
        return torch.stack(TP.aggregate_loop(mats, x), 0).transpose(0, 1)

    monkeypatch.setattr(ops, "core_aggregate", oracle_aggregate)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    n, T = 120, 3
    graphs = dynamic_graph(n, 6, T, seed=2)
    lists = [O.core_adj_list([O.kcore_matrices(g)], 0, 1, 1, max_core=3)[0] for g in graphs]
    adj = [CoreAdj.from_matrices(l) for l in lists]
    torch.manual_seed(0)
    model = ctgcn_amd.CTGCN(10, 12, 8, 1, 2, T)
    xs = [torch.randn(n, 10) for _ in range(T)]
    with torch.no_grad():
        want = model(xs, adj)                              # inference path
    for p in model.parameters():
        p.requires_grad_(False)
    xg = [x.clone().requires_grad_(True) for x in xs]
    assert model._needs_autograd(xg) and not model._needs_autograd(xs)
    out = model(xg, adj)
    assert out.requires_grad
    np.testing.assert_allclose(out.detach().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    gsel = torch.randn_like(want)              # (not out.square().sum(): LayerNorm makes that constant, its gradient is rounding noise)
    (out * gsel).sum().backward()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    xr = [x.clone().requires_grad_(True) for x in xs]
    ref = TP.ctgcn_with_grad(sd, xr, [[TP.coo_like_reference(m) for m in l] for l in lists])
    (ref * gsel).sum().backward()
    for a, b in zip(xg, xr):
        np.testing.assert_allclose(a.grad.numpy(), b.grad.numpy(), rtol=1e-4, atol=1e-5 * float(b.grad.abs().max()))


def test_core_adj_device_copies_are_cached():
    adj = CoreAdj.from_matrices([sp.eye(5, format="csr") + sp.csr_matrix(np.ones((5, 5)) - np.eye(5))])
    assert adj.to("cpu") is adj
    fake = adj._copy_to(torch.device("cpu"))
    adj._moved["meta"] = fake                              # what .to() stores after the first move to another device
    assert adj.to("meta") is fake


def test_columnar_edge_reader_equals_the_line_loop(tmp_path):
    """utils.read_edge_rows (Arrow CSV + hash join) returns exactly what the per-line loop returns — order, duplicates, self
    loops, float weights in every notation Python's float() accepts in these files, 2-column files — and raises KeyError for a
    name that is not in the node list, as the reference's dict lookup does (utils.py:52-53)."""
    from ctgcn_amd.utils import read_edge_rows, _read_edge_rows_loop
    rng = np.random.default_rng(4)
    names = ["N%d" % i for i in range(300)] + ["17", "0042", "a b"]
    node2idx = dict(zip(names, range(len(names))))
    rows = rng.integers(0, len(names), size=(5000, 2))
    weights = ["1", "2.5", "1e-3", "3.0000000000000004", "7E2", "-0.125", "12345678.875", "0.1"]
    path = tmp_path / "g.csv"
    with open(path, "w") as fp:
        fp.write("from_id\tto_id\tweight\n")
        for i, (a, b) in enumerate(rows):
            fp.write("%s\t%s\t%s\n" % (names[a], names[b], weights[i % len(weights)]))
    got, want = read_edge_rows(str(path), node2idx), _read_edge_rows_loop(str(path), node2idx)
    for g_, w_ in zip(got, want):
        assert g_.dtype == w_.dtype and np.array_equal(g_, w_)
    path2 = tmp_path / "g2.csv"
    with open(path2, "w") as fp:
        fp.write("from_id,to_id\n")
        for a, b in rows[:100]:
            fp.write("%s,%s\n" % (names[a], names[b]))
    got, want = read_edge_rows(str(path2), node2idx, sep=","), _read_edge_rows_loop(str(path2), node2idx, sep=",")
    for g_, w_ in zip(got, want):
        assert np.array_equal(g_, w_)
    (tmp_path / "empty.csv").write_text("from_id\tto_id\tweight\n")
    assert all(len(a) == 0 for a in read_edge_rows(str(tmp_path / "empty.csv"), node2idx))
    (tmp_path / "bad.csv").write_text("from_id\tto_id\tweight\nN1\tNOPE\t1\n")
    with pytest.raises(KeyError):
        read_edge_rows(str(tmp_path / "bad.csv"), node2idx)


def test_device_builder_for_nested_npz_lists_equals_the_host_builder():
    """CoreAdj.from_nested_matrices_device (the .npz loader route with has_cuda: K coordinate lists tagged with torch sort /
    unique ops on the loader's device) gives the same arrays as the host lexsort builder on the reference loader's outputs,
    and declines (None) lists that are not nested."""
    ca = load_golden("uci_core_adj.npz")
    for tag in ("mcm1_", "mc5_"):
        for t, k in enumerate(ca[tag + "K"]):
            mats = [csr_from(ca, tag + "t%d_j%d" % (t, j), 1899, np.float32) for j in range(int(k))]
            host = CoreAdj.from_matrices(mats)
            kept = [m.copy() for m in mats]
            first = kept[0].tolil()
            first.setdiag(0)                                  # the loader hands over A_kmax and asks for + I (helper.py:71-72)
            kept[0] = first.tocsr()
            kept[0].eliminate_zeros()
            dev = CoreAdj.from_nested_matrices_device(kept, "cpu", self_loop=True)
            assert dev is not None and dev.nested and dev.symmetric and dev.self_loop
            for a in ("row_ptr", "col", "val", "slot"):
                assert np.array_equal(getattr(host, a).numpy(), getattr(dev, a).numpy()), (tag, t, a)
            assert host.nnz_per_slot == dev.nnz_per_slot
    a = sp.random(50, 50, 0.1, format="csr", random_state=1)
    b = sp.random(50, 50, 0.1, format="csr", random_state=2)
    assert CoreAdj.from_nested_matrices_device([a + a.T, b + b.T], "cpu") is None


def test_row_plan_names_exactly_rows_that_repeat_in_the_oracle():
    """CoreAdj.row_plan (inference path): a clear bit j of a tile's mask promises that H[v, j] == H[v, j-1] for the 16 rows of
    the tile — checked against the C oracle's H = relu(cumulative A_k x) (layers.py:41-48) for a k-core list and for a general list;
    `order` is a permutation that keeps equal patterns together."""
    import scipy.sparse as sp
    from ctgcn_amd import CoreAdj
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O
    rng = np.random.default_rng(4)
    n = 1000
    src, dst = rng.integers(0, n, 450), rng.integers(0, n, 450)
    hub = rng.choice(n, 30, replace=False)
    src, dst = np.concatenate([src, rng.choice(hub, 400)]), np.concatenate([dst, rng.choice(hub, 400)])
    csr = symmetric_csr_from_rows(src, dst, np.ones(len(src)), n)
    nested = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=5)[0]
    general = [sp.random(n, n, density=0.001 * (1 + j % 2), random_state=j, format="csr", dtype=np.float32) for j in range(4)]
    x = rng.standard_normal((n, 8)).astype(np.float32)
    wide = [sp.random(n, n, density=0.0004, random_state=50 + j, format="csr", dtype=np.float32) for j in range(32)]      # K = 32: bit 31 of the masks
    for mats, kw in ((nested, {}), (general, dict(self_loop=False)), (wide, dict(self_loop=False))):
        adj = CoreAdj.from_matrices(mats, **kw)
        plan = adj.row_plan()
        K = adj.K
        order, tmask = plan["order"].numpy().astype(np.int64), plan["tile_mask"].numpy().astype(np.int64)
        assert np.array_equal(np.sort(order), np.arange(n)) and np.array_equal(plan["inverse"].numpy()[order], np.arange(n))
        assert len(tmask) == -(-n // 16) and np.all(tmask & 1)
        H = O.core_aggregate(mats, x)                       # [n, K, d], ReLU applied
        repeats = np.zeros((n, K), dtype=bool)
        repeats[:, 1:] = (H[:, 1:] == H[:, :-1]).all(axis=2)
        bits = (tmask[np.arange(n) // 16][:, None] >> np.arange(K)[None, :]) & 1          # [position, slot]
        assert repeats[order][bits == 0].all(), "the plan skips a row that is not a repeat"
        # patterns are grouped: the masks are not looser than the rows' own patterns except at group boundaries (< 2^K tiles)
        own = ~repeats[order]
        loose = ((bits == 1) & ~own).any(axis=1).reshape(-1)
        if K <= 8:                                            # (a list with 2^32 possible patterns groups nothing: only correctness matters there)
            assert loose.sum() <= 16 * (1 << K), loose.sum()
        assert plan["new_rows"] == 16 * sum(bin(int(m) & ((1 << K) - 1)).count("1") for m in tmask)
        # the GEMM consumer's compact layout (tiles of 64): every wanted (row, slot) has its own operand row inside its tile's range
        p64 = adj.row_plan(adj.PLAN_TILE_GEMM)
        tb, tm64 = p64["tile_base"].numpy().astype(np.int64), p64["tile_mask"].numpy().astype(np.int64) & 0xffffffff
        cnt = np.array([bin(int(m)).count("1") for m in tm64])
        assert np.array_equal(tb, np.cumsum(cnt * 64) - cnt * 64) and p64["operand_rows"] == int((cnt * 64).sum())
        dest = adj.plan_row_dest(p64, torch.arange(n), True).numpy().reshape(n, K)
        pos = p64["inverse"].numpy().astype(np.int64)
        want_bits = (tm64[pos // 64][:, None] >> np.arange(K)[None, :]) & 1
        assert np.array_equal(dest >= 0, want_bits == 1)
        used = dest[dest >= 0]
        assert len(np.unique(used)) == len(used) and used.max() < p64["operand_rows"]
        lo, hi = tb[pos // 64], tb[pos // 64] + cnt[pos // 64] * 64
        assert ((dest >= lo[:, None]) | (dest < 0)).all() and (dest < hi[:, None]).all()
        assert np.array_equal(adj.plan_row_dest(plan, torch.arange(n), False).numpy().reshape(n, K),
                              np.where(bits[np.argsort(order)] == 1, (np.argsort(order) * K)[:, None] + np.arange(K)[None, :], -1))
    assert CoreAdj.from_matrices([general[0]] * 1, self_loop=True).row_plan()["new_rows"] == -(-n // 16) * 16


def test_row_plan_over_more_than_32_slots():
    """33-64 slots: two 32-bit mask words per tile (word 2 T = slots 0-31, word 2 T + 1 = slots 32-63); popcounts, bases and the hub-row
    destinations follow the 64-bit pattern; beyond 64 slots there is no plan."""
    import scipy.sparse as sp
    from ctgcn_amd import CoreAdj
    n, K = 70, 40
    mats = [sp.random(n, n, density=0.01, random_state=j, format="csr", dtype=np.float32) for j in range(K)]
    adj = CoreAdj.from_matrices(mats, self_loop=False)
    for tile in (adj.PLAN_TILE, adj.PLAN_TILE_GEMM):
        plan = adj.row_plan(tile)
        nt = -(-n // tile)
        words = plan["tile_mask"].long() & 0xffffffff
        assert words.numel() == 2 * nt and bool((words[1::2] < (1 << (K - 32))).all())
        order = plan["order"].long().numpy()
        has = np.zeros((n, K), bool)
        for j, m in enumerate(mats):
            has[np.unique(m.nonzero()[0]), j] = True
        has[:, 0] = True
        total = 0
        for t in range(nt):
            rows = order[t * tile:(t + 1) * tile]
            want = has[rows].any(0)
            got = np.array([(int(words[2 * t + (j >> 5)]) >> (j & 31)) & 1 for j in range(K)], bool)
            assert (got == want).all(), (tile, t)
            assert int(plan["tile_base"][t]) == total
            total += int(want.sum()) * tile
        assert plan["operand_rows"] == total
        dest = adj.plan_row_dest(plan, torch.arange(n), compact=True).view(n, K).numpy()
        inv = plan["inverse"].long().numpy()
        for v in (0, n // 2, n - 1):
            t = inv[v] // tile
            bits = has[order[t * tile:(t + 1) * tile]].any(0)
            rank = np.cumsum(bits) - bits
            want = np.where(bits, int(plan["tile_base"][t]) + (inv[v] % tile) * bits.sum() + rank, -1)
            assert (dest[v] == want).all()
    assert CoreAdj.from_matrices(mats + mats[:25], self_loop=False).row_plan() is None      # 65 slots
