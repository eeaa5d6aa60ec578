"""Pins the CPU oracle (oracle/) to vectors produced by the reference itself (tests/golden/)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_golden, csr_from, close_scaled, formula_tensor
from oracle import oracle as O
from oracle import torch_path as TP


def _same_csr(a, b, exact_values=True):
    a, b = sp.csr_matrix(a), sp.csr_matrix(b)
    a.sort_indices(); b.sort_indices()
    assert a.shape == b.shape and a.nnz == b.nnz
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.data.astype(np.float64), b.data.astype(np.float64))


# --------------------------------------------------------------------------- k-core numbers
def test_toy_kcore_known_answers():
    g = load_golden("toy_kcore.npz")
    for name in g["names"]:
        n = int(g[name + "_n"])
        e = g[name + "_edges"]
        adj = O.adjacency_from_edge_rows(e[:, 0], e[:, 1], np.ones(len(e)), n)
        assert np.array_equal(O.core_numbers(adj), g[name + "_core"]), name
    # the reference's own toy graph shape: 4-clique + pendant, separate edge
    assert g["clique_path_core"].tolist() == [3, 3, 3, 3, 1, 1, 1]


def test_uci_core_numbers_and_kcore_files_bit_exact():
    snaps = load_golden("uci_snapshots.npz")
    kc = load_golden("uci_kcore.npz")
    n = len(snaps["node_names"])
    maxcores = []
    for t in range(len(snaps["files"])):
        adj = O.adjacency_from_edge_rows(snaps["t%d_src" % t], snaps["t%d_dst" % t], snaps["t%d_w" % t], n)
        core = O.core_numbers(adj)
        assert np.array_equal(core, kc["core_t%d" % t])
        mats = O.kcore_matrices(adj, core)
        files = [str(f) for f in kc["t%d_files" % t]]
        assert files == O.core_file_names(len(mats))
        maxcores.append(len(mats))
        for f, m in zip(files, mats):
            _same_csr(m, csr_from(kc, "t%d_%s" % (t, f[:-4]), n))
    assert maxcores == [8, 16, 6, 5, 4, 3, 2]          # SURVEY.md §8a row a10 [probe]


def _uci_mats():
    kc = load_golden("uci_kcore.npz")
    n = len(kc["core_t0"])
    return [[csr_from(kc, "t%d_%s" % (t, str(f)[:-4]), n) for f in kc["t%d_files" % t]] for t in range(7)], n


@pytest.mark.parametrize("tag,start,dur,mc,expectK", [
    ("mcm1_", 0, 7, -1, [8, 8, 6, 5, 4, 3, 2]),       # quirk A: sticky max_core from snapshot 0
    ("mc5_", 0, 7, 5, [5, 5, 5, 5, 4, 3, 2]),
    ("w4_", 4, 3, -1, [4, 3, 2]),
])
def test_uci_loader_semantics(tag, start, dur, mc, expectK):
    mats, n = _uci_mats()
    ca = load_golden("uci_core_adj.npz")
    got = O.core_adj_list(mats, start, dur, 7, max_core=mc)
    assert [len(g) for g in got] == expectK == ca[tag + "K"].tolist()
    for t, inner in enumerate(got):
        for j, m in enumerate(inner):
            _same_csr(m, csr_from(ca, tag + "t%d_j%d" % (t, j), n))
    if tag == "mcm1_":
        assert [m.nnz for m in got[0]] == [2677, 1506, 1872, 2208, 2500, 2830, 3236, 3542]
    if tag == "mc5_":
        assert [m.nnz for m in got[0]] == [4107, 2500, 2830, 3236, 3542]


# ------------------------------------------------------------- weighted graphs, duplicates etc.
def test_weighted_small_pipeline_and_aggregation():
    g = load_golden("weighted_small.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        n = int(g[p + "n"])
        per_snap = []
        for s in range(2):
            adj = O.adjacency_from_edge_rows(g[p + "s%d_src" % s], g[p + "s%d_dst" % s], g[p + "s%d_w" % s], n)
            _same_csr(adj, csr_from(g, p + "s%d_dateadj" % s, n))           # utils.get_sp_adj_mat
            core = O.core_numbers(adj)
            assert np.array_equal(core, g[p + "s%d_core" % s])
            mats = O.kcore_matrices(adj, core)
            files = [str(f) for f in g[p + "core_t%d_files" % s]]
            assert files == O.core_file_names(len(mats))
            for f, m in zip(files, mats):
                _same_csr(m, csr_from(g, p + "core_t%d_%s" % (s, f[:-4]), n))
            per_snap.append(mats)
        adj_list = O.core_adj_list(per_snap, 0, 2, 2, max_core=int(g[p + "max_core"]))
        assert [len(a) for a in adj_list] == g[p + "adj_K"].tolist()
        for t, inner in enumerate(adj_list):
            for j, m in enumerate(inner):
                _same_csr(m, csr_from(g, p + "adj_t%d_j%d" % (t, j), n))
        # aggregation stage (layers.py:41-48,58) and its gradient
        x = g[p + "cd_x"]
        H = O.core_aggregate(adj_list[1], x)
        close_scaled(H, g[p + "cd_agg"])
        # full layer through the torch restatement, forward and dX
        sd = {k[len(p + "cd_sd_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(p + "cd_sd_")}
        xt = torch.from_numpy(x).requires_grad_(True)
        tadj = [TP.coo_like_reference(m) for m in adj_list[1]]
        out = TP.core_diffusion(sd, "", xt, tadj, str(g[p + "cd_rnn"]))
        np.testing.assert_allclose(out.detach().numpy(), g[p + "cd_out"], rtol=1e-4, atol=1e-5)
        (out * torch.from_numpy(g[p + "cd_gout"])).sum().backward()
        np.testing.assert_allclose(xt.grad.numpy(), g[p + "cd_dx"], rtol=1e-4, atol=2e-5)


def test_aggregate_backward_matches_autograd():
    g = load_golden("weighted_small.npz")
    p = "c1_"
    n = int(g[p + "n"])
    adj = [csr_from(g, p + "adj_t1_j%d" % j, n, np.float32) for j in range(int(g[p + "adj_K"][1]))]
    x = torch.from_numpy(g[p + "cd_x"]).requires_grad_(True)
    hs = TP.aggregate_loop([TP.coo_like_reference(m) for m in adj], x)
    H = torch.stack(hs, 0).transpose(0, 1)
    dH = torch.from_numpy(formula_tensor(tuple(H.shape), 0.7, 0.2))
    (H * dH).sum().backward()
    close_scaled(O.core_aggregate_bwd(adj, g[p + "cd_x"], dH.numpy()), x.grad.numpy(), rtol=2e-5, atol_scale=4e-6)


# ----------------------------------------------------------------------------- whole models
def _sd(g, tag):
    return {k[len(tag + "sd_"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "sd_")}


def _uci_window(start, dur):
    mats, n = _uci_mats()
    adj = O.core_adj_list(mats, start, dur, 7, max_core=-1)
    return [[TP.coo_like_reference(m) for m in inner] for inner in adj], n


def test_models_match_reference_outputs():
    g = load_golden("models_uci.npz")
    start, dur = int(g["start"]), int(g["duration"])
    adj, n = _uci_window(start, dur)
    eye = [TP.coo_like_reference(sp.eye(n, format="csr")) for _ in range(dur)]
    xd = [torch.from_numpy(a) for a in formula_tensor((dur, n, 24), 0.11, 0.3)]
    tol = dict(rtol=1e-4, atol=2e-5)

    out = TP.ctgcn(_sd(g, "ctgcn_c_"), eye, adj, "GRU", "C", "L")
    np.testing.assert_allclose(out.numpy(), g["ctgcn_c_out"], **tol)
    out, tr = TP.ctgcn(_sd(g, "ctgcn_s_"), xd, adj, "GRU", "S", "N")
    np.testing.assert_allclose(out.numpy(), g["ctgcn_s_out"], **tol)
    np.testing.assert_allclose(torch.stack(tr).numpy(), g["ctgcn_s_trans"], **tol)
    out = TP.ctgcn(_sd(g, "ctgcn_c_lstm_"), xd, adj, "LSTM", "C", "L")
    np.testing.assert_allclose(out.numpy(), g["ctgcn_c_lstm_out"], **tol)
    out = TP.cgcn(_sd(g, "cgcn_c_"), xd, adj, "GRU", "C", "L")
    np.testing.assert_allclose(torch.stack(out).numpy(), g["cgcn_c_out"], **tol)
    out, tr = TP.cgcn(_sd(g, "cgcn_s_"), xd, adj, "GRU", "S", "N")
    np.testing.assert_allclose(torch.stack(out).numpy(), g["cgcn_s_out"], **tol)
    np.testing.assert_allclose(torch.stack(tr).numpy(), g["cgcn_s_trans"], **tol)
    out = TP.cgcn(_sd(g, "cgcn_c_single_"), xd[0], adj[0], "GRU", "C", "N")
    np.testing.assert_allclose(out.numpy(), g["cgcn_c_single_out"], **tol)


def test_width_128_models_match_reference_outputs():
    """hidden = embed = 128 (the width of every shipped config): reference CoreDiffusion / CTGCN-C / CTGCN-S outputs and
    gradients (tests/golden/models_w128.npz) vs the torch restatement, weights regenerated from the seeded-numpy helper."""
    from conftest import seeded_parameters, check_sampled_tensor
    g = load_golden("models_w128.npz")
    rows = g["rows"]
    adj, n = _uci_window(4, 3)
    dur = 3

    class _Bag(torch.nn.Module):          # named parameters in a module the helper can visit; names = reference state_dict keys
        def __init__(self, shapes):
            super().__init__()
            self.names = sorted(shapes)
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(shapes[k])) for k in self.names])

        def named_parameters(self, *a, **k):
            return iter(zip(self.names, self.ps))

    def shapes_cd(prefix, din, dout):
        return {prefix + "linear.weight": (dout, din), prefix + "linear.bias": (dout,),
                prefix + "rnn.weight_ih_l0": (3 * dout, din), prefix + "rnn.weight_hh_l0": (3 * dout, dout),
                prefix + "rnn.bias_ih_l0": (3 * dout,), prefix + "rnn.bias_hh_l0": (3 * dout,),
                prefix + "norm.weight": (dout,), prefix + "norm.bias": (dout,)}

    # (1) the layer
    bag = _Bag(shapes_cd("", 128, 128))
    seeded_parameters(bag, 11)
    sd = dict(bag.named_parameters())
    x = torch.from_numpy(formula_tensor((n, 128), 0.19, 0.2)).requires_grad_(True)
    gout = torch.from_numpy(formula_tensor((n, 128), 0.41, 0.9))
    saved = TP._rnn
    TP._rnn = TP._rnn_grad
    try:
        out = TP.core_diffusion(sd, "", x, adj[0], "GRU")
    finally:
        TP._rnn = saved
    np.testing.assert_allclose(out.detach().numpy()[rows], g["cd_out_rows"], rtol=1e-4, atol=1e-5)
    (out * gout).sum().backward()
    check_sampled_tensor(g, "cd_out", out.detach().numpy(), 1e-4, 1e-5)
    np.testing.assert_allclose(x.grad.numpy()[rows], g["cd_dx_rows"], rtol=1e-4, atol=1e-5 * float(np.abs(g["cd_dx_rows"]).max()))
    for name, p in sd.items():
        if "linear" in name:
            continue
        check_sampled_tensor(g, "cd_grad_" + name, p.grad.numpy(), 2e-4, 2e-5)

    # (2) CTGCN-C and CTGCN-S
    xd = [torch.from_numpy(a) for a in formula_tensor((dur, n, 24), 0.11, 0.3)]
    gsel = torch.from_numpy(formula_tensor((dur, n, 128), 0.37, 1.1))
    for tag, seed, mtype, act, trans_num, diff_num in (("ctgcn_c_", 21, "C", "L", 1, 2), ("ctgcn_s_", 22, "S", "N", 3, 1)):
        shapes = {}
        for t in range(dur):
            if trans_num == 1:
                shapes["mlp_list.%d.linear.weight" % t] = (128, 24)
                shapes["mlp_list.%d.linear.bias" % t] = (128,)
            else:
                widths = [24] + [128] * trans_num
                for i in range(trans_num):
                    shapes["mlp_list.%d.linears.%d.weight" % (t, i)] = (widths[i + 1], widths[i])
                    shapes["mlp_list.%d.linears.%d.bias" % (t, i)] = (widths[i + 1],)
            for l in range(diff_num):
                shapes.update(shapes_cd("duffision_list.%d.diffusion_list.%d." % (t, l), 128, 128))
        shapes.update({"rnn.weight_ih_l0": (384, 128), "rnn.weight_hh_l0": (384, 128), "rnn.bias_ih_l0": (384,),
                       "rnn.bias_hh_l0": (384,), "norm.weight": (128,), "norm.bias": (128,)})
        bag = _Bag(shapes)
        seeded_parameters(bag, seed)
        sd = dict(bag.named_parameters())
        res = TP.ctgcn_with_grad(sd, xd, adj, "GRU", mtype, act)
        if mtype == "S":
            res, trans = res
            np.testing.assert_allclose(torch.stack(trans).detach().numpy()[:, rows], g[tag + "trans_rows"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(res.detach().numpy()[:, rows], g[tag + "out_rows"], rtol=1e-4, atol=1e-5)
        check_sampled_tensor(g, tag + "out", res.detach().numpy(), 1e-4, 1e-5)
        (res * gsel).sum().backward()
        for name, p in sd.items():
            if ".linear." in name and "mlp_list" not in name:
                continue                       # CoreDiffusion.linear is unused (layers.py:24): zero gradient on both sides
            check_sampled_tensor(g, tag + "grad_" + name, p.grad.numpy(), 5e-4, 5e-5)


# ------------------------------------------------------------------ random walks / negative sampling (§8f rank 3)
def test_walk_corpus_oracle_matches_reference_on_deterministic_graph():
    g = load_golden("negloss.npz")
    n = len(g["match_adj_indptr"]) - 1
    adj = csr_from(g, "match_adj", n)
    L, W = [int(x) for x in g["match_LW"]]
    pairs, freq = O.matching_walk_outputs(adj, L, W)
    _same_csr(pairs, csr_from(g, "match_pairs", n))
    assert np.array_equal(O.negative_table(freq), g["match_neg"])


def test_neg_sampling_loss_oracle_matches_reference():
    g = load_golden("negloss.npz")
    n2, T, dim, neg_num, Q = [int(x) for x in g["loss_cfg"]]
    batch = g["loss_batch"]
    embs, ni, pi, gi = [], [], [], []
    for t in range(T):
        lens, flat = g["loss_pairs%d_len" % t], g["loss_pairs%d_flat" % t]
        ptr = np.concatenate([[0], np.cumsum(lens)])
        nodes, pos = [], []
        for b in batch:
            part = flat[ptr[b]:ptr[b + 1]]
            nodes += [b] * len(part); pos += part.tolist()
        embs.append(torch.from_numpy(g["loss_emb%d" % t]).requires_grad_(True))
        ni.append(torch.tensor(nodes, dtype=torch.int64)); pi.append(torch.tensor(pos, dtype=torch.int64))
        gi.append(torch.from_numpy(g["loss_table%d" % t]))
    loss = O.neg_sampling_loss(embs, ni, pi, gi, Q)
    np.testing.assert_allclose(loss.detach().numpy(), g["loss_value"], rtol=1e-5)
    loss.backward()
    for t in range(T):
        np.testing.assert_allclose(embs[t].grad.numpy(), g["loss_grad%d" % t], rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------ the row-subset form of the oracle (full-size GPU parity tests)
def test_row_subset_path_equals_the_full_path():
    """TP.ctgcn_rows / TP.cdn_rows (the reference path evaluated for some rows only: sliced matrices, nothing else) against TP.ctgcn /
    TP.ctgcn_with_grad — themselves pinned to the reference's outputs above — on whole graphs: outputs of the rows, and every gradient of
    a loss that reads those rows only, in float64 (agreement to rounding) and float32."""
    from ctgcn_amd.synth import dynamic_graph
    n, T = 400, 3
    graphs = dynamic_graph(n, avg_deg=6, snapshots=T, seed=5)
    mats = O.core_adj_list([O.kcore_matrices(g) for g in graphs], 0, T, T, max_core=4)
    assert min(len(m) for m in mats) >= 2
    adj = [[TP.coo_like_reference(m) for m in l] for l in mats]
    torch.manual_seed(3)
    import ctgcn_amd
    model = ctgcn_amd.CTGCN(n, 24, 16, 1, 3, T)             # three CoreDiffusion layers: two levels of column sets
    idx = torch.arange(n).repeat(2, 1)
    rows = np.sort(np.random.default_rng(1).choice(n, 37, replace=False))
    G = torch.randn(T, len(rows), 16, dtype=torch.float64)
    for dtype, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        xs = [torch.sparse_coo_tensor(idx, torch.ones(n, dtype=dtype), (n, n)) for _ in range(T)]
        sd_a = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in model.state_dict().items()}
        sd_b = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in model.state_dict().items()}
        full = TP.ctgcn_with_grad(sd_a, xs, [[a.to(dtype) for a in l] for l in adj])
        part = TP.ctgcn_rows(sd_b, xs, mats, rows, with_grad=True)
        assert part.shape == (T, len(rows), 16)
        assert (full[:, rows] - part).abs().max().item() <= tol
        (full[:, rows] * G.to(dtype)).sum().backward()
        (part * G.to(dtype)).sum().backward()
        for k in sd_a:
            if sd_a[k].grad is None:
                assert sd_b[k].grad is None or sd_b[k].grad.abs().max().item() == 0.0, k
                continue
            scale = sd_a[k].grad.abs().max().item()
            assert (sd_a[k].grad - sd_b[k].grad).abs().max().item() <= tol * max(scale, 1.0) * 10, k
        with torch.no_grad():                               # the inference form (nn.GRU) of the same rows
            part_inf = TP.ctgcn_rows({k: v.detach() for k, v in sd_b.items()}, xs, mats, rows)
        assert (part_inf - part.detach()).abs().max().item() <= max(tol, 1e-6) * 10


def test_row_subset_path_with_dense_features_equals_the_full_path():
    """TP.ctgcn_rows on DENSE features (the CTGCN-S input) transforms only the rows the diffusion layers read: same outputs and gradients as the
    full path, float64 to rounding."""
    from ctgcn_amd.synth import dynamic_graph
    import ctgcn_amd
    n, T = 300, 3
    graphs = dynamic_graph(n, avg_deg=5, snapshots=T, seed=8)
    mats = O.core_adj_list([O.kcore_matrices(g) for g in graphs], 0, T, T, max_core=3)
    adj = [[TP.coo_like_reference(m).double() for m in l] for l in mats]
    torch.manual_seed(4)
    model = ctgcn_amd.CTGCN(11, 20, 16, 3, 2, T, model_type="S", trans_activate_type="N")
    xs = [torch.randn(n, 11, dtype=torch.float64) for _ in range(T)]
    rows = np.sort(np.random.default_rng(2).choice(n, 29, replace=False))
    G = torch.randn(T, len(rows), 16, dtype=torch.float64)
    sd_a = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    sd_b = {k: v.detach().double().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    full, _ = TP.ctgcn_with_grad(sd_a, xs, adj, "GRU", "S", "N")
    part = TP.ctgcn_rows(sd_b, xs, mats, rows, "GRU", "S", "N", with_grad=True)
    assert (full[:, rows] - part).abs().max().item() <= 1e-12
    (full[:, rows] * G).sum().backward()
    (part * G).sum().backward()
    compared = 0
    for k in sd_a:
        if sd_a[k].grad is None:
            assert sd_b[k].grad is None or sd_b[k].grad.abs().max().item() == 0.0, k
            continue
        assert (sd_a[k].grad - sd_b[k].grad).abs().max().item() <= 1e-11 * max(1.0, sd_a[k].grad.abs().max().item()), k
        compared += 1
    assert compared >= 20

