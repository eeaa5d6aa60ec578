"""HIP path vs the CPU oracle / golden vectors, through the C ABI (ctypes) — needs an MI355X."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import load_golden, csr_from, close_scaled

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _upload(csr):
    csr = sp.csr_matrix(csr)
    csr.sort_indices()
    d = _dev()
    return (torch.from_numpy(csr.indptr.astype(np.int32)).to(d), torch.from_numpy(csr.indices.astype(np.int32)).to(d),
            torch.from_numpy(csr.data.astype(np.float32)).to(d))


def _rand_graph(n, m, seed, weighted=True):
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    p = 1.0 / (np.arange(n) + 5.0) ** 0.7
    p /= p.sum()
    src, dst = rng.choice(n, m, p=p), rng.choice(n, m, p=p)
    w = rng.integers(1, 9, m) * 0.25 if weighted else np.ones(m)
    from ctgcn_amd.utils import symmetric_csr_from_rows
    return symmetric_csr_from_rows(src, dst, w, n)


# ------------------------------------------------------------------------------------------ k-core
def test_library_loaded_and_device():
    from ctgcn_amd import _lib
    import ctypes
    lib = _lib.load()
    name = ctypes.create_string_buffer(128)
    cus = ctypes.c_int(0)
    assert lib.ctgcn_device_info(name, 128, ctypes.byref(cus)) == 0
    assert b"gfx950" in name.value, name.value
    assert cus.value == 256


def test_kcore_toy_and_uci_bit_exact():
    from ctgcn_amd import ops
    from ctgcn_amd.utils import symmetric_csr_from_rows
    g = load_golden("toy_kcore.npz")
    for name in g["names"]:
        n, e = int(g[name + "_n"]), g[name + "_edges"]
        csr = symmetric_csr_from_rows(e[:, 0], e[:, 1], np.ones(len(e)), n) if len(e) else sp.csr_matrix((n, n))
        rp, col, _ = _upload(csr)
        core, mx = ops.kcore(rp, col)
        assert np.array_equal(core.cpu().numpy(), g[name + "_core"]), name
        assert mx == int(g[name + "_core"].max())
    snaps, kc = load_golden("uci_snapshots.npz"), load_golden("uci_kcore.npz")
    n = len(snaps["node_names"])
    for t in range(7):
        csr = symmetric_csr_from_rows(snaps["t%d_src" % t], snaps["t%d_dst" % t], snaps["t%d_w" % t], n)
        rp, col, _ = _upload(csr)
        core, mx = ops.kcore(rp, col)
        assert np.array_equal(core.cpu().numpy(), kc["core_t%d" % t])


@pytest.mark.parametrize("n,m,seed", [(1, 0, 0), (64, 0, 1), (5000, 40000, 2), (200000, 1600000, 3), (3000, 400000, 4)])
def test_kcore_random_vs_oracle(n, m, seed):
    from ctgcn_amd import ops
    from oracle import oracle as O
    csr = _rand_graph(n, m, seed) if m else sp.csr_matrix((n, n))
    rp, col, _ = _upload(csr)
    core, mx = ops.kcore(rp, col)
    ref = O.core_numbers(csr)
    assert np.array_equal(core.cpu().numpy(), ref)
    assert mx == int(ref.max(initial=0))


def test_kcore_level_cap_reports_min_of_core_and_cap():
    from ctgcn_amd import ops
    from oracle import oracle as O
    csr = _rand_graph(50000, 600000, 13)
    rp, col, _ = _upload(csr)
    ref = O.core_numbers(csr)
    for cap in (1, 2, 5, int(ref.max()), int(ref.max()) + 7):
        core, mx = ops.kcore(rp, col, level_cap=cap)
        assert np.array_equal(core.cpu().numpy(), np.minimum(ref, cap)), cap
        assert mx == min(int(ref.max()), cap)


def test_kcore_long_path_and_queue_spill():
    """a 300k-vertex path (every vertex peels at level 1 through a chain of pushes) and a graph whose
    level-k frontier exceeds one block's LDS queue."""
    from ctgcn_amd import ops
    from oracle import oracle as O
    n = 300000
    a = np.arange(n - 1)
    csr = sp.coo_matrix((np.ones(2 * (n - 1)), (np.concatenate([a, a + 1]), np.concatenate([a + 1, a]))), shape=(n, n)).tocsr()
    rp, col, _ = _upload(csr)
    core, mx = ops.kcore(rp, col)
    assert mx == 1 and bool((core == 1).all())
    # star forest: 40 hubs x 20000 leaves; all leaves of a hub sit in few blocks' ranges -> hub chase spills
    hubs, leaves = 40, 20000
    h = np.repeat(np.arange(hubs), leaves)
    l = hubs + np.arange(hubs * leaves)
    n2 = hubs + hubs * leaves
    csr = sp.coo_matrix((np.ones(2 * len(h)), (np.concatenate([h, l]), np.concatenate([l, h]))), shape=(n2, n2)).tocsr()
    rp, col, _ = _upload(csr)
    core, mx = ops.kcore(rp, col)
    assert np.array_equal(core.cpu().numpy(), O.core_numbers(csr))


# --------------------------------------------------------------------------------- aggregation
def _agg_case(csr_list, d, seed, self_loop=None, check_bwd=True, rtol=1e-5):
    from ctgcn_amd import ops, CoreAdj
    from oracle import oracle as O
    n = csr_list[0].shape[0]
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    adj = CoreAdj.from_matrices(csr_list, self_loop=self_loop, device=_dev())
    ref_list = adj.to_scipy_list()             # reference-semantics matrices (incl. + I)
    xt = torch.from_numpy(x).to(_dev()).requires_grad_(True)
    H = ops.core_aggregate(xt, adj, relu=True)
    Href = O.core_aggregate(ref_list, x)
    close_scaled(H.detach().cpu().numpy(), Href, rtol=rtol)
    if check_bwd:
        dH = rng.standard_normal(Href.shape).astype(np.float32)
        (H * torch.from_numpy(dH).to(_dev())).sum().backward()
        close_scaled(xt.grad.cpu().numpy(), O.core_aggregate_bwd(ref_list, x, dH), rtol=2e-5, atol_scale=4e-6)
    return adj


def test_aggregate_golden_small():
    from ctgcn_amd import ops, CoreAdj
    g = load_golden("weighted_small.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        n = int(g[p + "n"])
        mats = [csr_from(g, p + "adj_t1_j%d" % j, n, np.float32) for j in range(int(g[p + "adj_K"][1]))]
        adj = CoreAdj.from_matrices(mats, device=_dev())
        assert adj.nested and adj.self_loop and adj.symmetric
        x = torch.from_numpy(g[p + "cd_x"]).to(_dev())
        H = ops.core_aggregate(x, adj)
        close_scaled(H.cpu().numpy(), g[p + "cd_agg"])


@pytest.mark.parametrize("d", [1, 3, 8, 12, 64, 128, 130, 256, 500, 1028])
def test_aggregate_feature_widths(d):
    from oracle import oracle as O
    csr = _rand_graph(700, 6000, 11)
    mats = O.kcore_matrices(csr)
    kept = O.core_adj_list([mats], 0, 1, 1, max_core=-1)[0]
    adj = _agg_case(kept, d, seed=d)
    assert adj.nested and adj.self_loop


def test_aggregate_not_nested_and_asymmetric():
    """arbitrary adjacency lists: overlapping but non-nested matrices, directed (asymmetric) weights."""
    rng = np.random.default_rng(5)
    n = 300
    mats = []
    for j in range(4):
        m = sp.random(n, n, density=0.03, random_state=100 + j, format="csr", dtype=np.float32)
        m.data = rng.standard_normal(m.nnz).astype(np.float32)
        mats.append(m)
    adj = _agg_case(mats, 40, seed=3, self_loop=False)
    assert not adj.nested and not adj.symmetric
    # nested but asymmetric: directed graph's sub-matrices
    a = mats[0]
    b = (a + mats[1]).tocsr()
    adj = _agg_case([a, b.multiply(b != 0).tocsr()], 24, seed=4, self_loop=False)


def test_aggregate_hub_rows_and_empty_rows():
    from oracle import oracle as O
    n = 5000
    rng = np.random.default_rng(9)
    hub = np.zeros(3000, dtype=np.int64)
    other = rng.integers(1, n - 100, 3000)          # last 100 vertices isolated
    src = np.concatenate([hub, rng.integers(1, n - 100, 20000)])
    dst = np.concatenate([other, rng.integers(1, n - 100, 20000)])
    from ctgcn_amd.utils import symmetric_csr_from_rows
    csr = symmetric_csr_from_rows(src, dst, np.ones(len(src)), n)
    kept = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=4)[0]
    _agg_case(kept, 128, seed=1)


def test_aggregate_deterministic_bitwise():
    from ctgcn_amd import ops, CoreAdj
    from oracle import oracle as O
    csr = _rand_graph(20000, 300000, 21)
    kept = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=6)[0]
    adj = CoreAdj.from_matrices(kept, device=_dev())
    x = torch.randn(20000, 128, device=_dev())
    a = ops.core_aggregate(x, adj)
    b = ops.core_aggregate(x, adj)
    assert torch.equal(a, b)


def test_spmm_csr_matches_torch_sparse_and_accumulates():
    from ctgcn_amd import ops
    csr = _rand_graph(4000, 50000, 31)
    rp, col, val = _upload(csr)
    for d in (128, 50):
        x = torch.randn(4000, d, device=_dev())
        y = ops.spmm_csr(rp, col, val, x)
        coo = csr.tocoo()
        ref = torch.sparse.mm(torch.sparse_coo_tensor(np.vstack((coo.row, coo.col)), coo.data.astype(np.float32), coo.shape), x.cpu())
        close_scaled(y.cpu().numpy(), ref.numpy())
        y2 = ops.spmm_csr(rp, col, val, x, out=y.clone(), accumulate=True)
        close_scaled(y2.cpu().numpy(), 2 * ref.numpy())


def test_abi_rejects_bad_arguments():
    from ctgcn_amd import _lib
    import ctypes
    lib = _lib.load()
    x = torch.zeros(4, 4, device=_dev())
    rc = lib.ctgcn_core_aggregate_f32(4, 4, 0, None, None, None, None, x.data_ptr(), 4, x.data_ptr(), 0, None, 0, 0, 1, None, 0, None)
    assert rc == -1 and b"K=0" in lib.ctgcn_last_error()
    rc = lib.ctgcn_kcore_i32(4, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 8, -1, None, None)
    assert rc == -3
    with pytest.raises(_lib.CtgcnHipError):
        _lib.check(rc, "kcore")


# --------------------------------------------------------------- native builder == file route
def test_native_route_equals_matrix_route():
    from ctgcn_amd import CoreAdj
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O
    snaps = load_golden("uci_snapshots.npz")
    n = len(snaps["node_names"])
    graphs = [symmetric_csr_from_rows(snaps["t%d_src" % t], snaps["t%d_dst" % t], snaps["t%d_w" % t], n) for t in range(7)]
    g = load_golden("weighted_small.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        graphs.append(symmetric_csr_from_rows(g[p + "s1_src"], g[p + "s1_dst"], g[p + "s1_w"], int(g[p + "n"])))
    graphs.append(_rand_graph(30000, 400000, 77))
    for gi, csr in enumerate(graphs):
        for mc in (-1, 3):
            adj, core, files = core_adj_from_scipy(csr, mc, _dev())
            mats = O.kcore_matrices(csr)
            assert files == (len(mats) if mc < 0 else min(len(mats), mc))          # capped peel reports min(max core, max_core)
            want_core = O.core_numbers(csr)
            assert np.array_equal(core.cpu().numpy(), want_core if mc < 0 else np.minimum(want_core, mc))
            ref = CoreAdj.from_matrices(O.core_adj_list([mats], 0, 1, 1, max_core=mc)[0], device="cpu")
            assert (adj.K, adj.nested, adj.self_loop, adj.symmetric) == (ref.K, True, True, True)
            assert adj.nnz_per_slot == ref.nnz_per_slot, (gi, mc)
            for a, b in ((adj.row_ptr, ref.row_ptr), (adj.col, ref.col), (adj.val, ref.val), (adj.slot, ref.slot)):
                assert torch.equal(a.cpu(), b), (gi, mc)


# ------------------------------------------------------------------------ edge rows -> CSR on the GPU
def _ingest_case(src, dst, w, n):
    from ctgcn_amd import ops
    from oracle import oracle as O
    d = _dev()
    rp, col, val = ops.edges_to_csr(torch.from_numpy(np.asarray(src, np.int32)).to(d), torch.from_numpy(np.asarray(dst, np.int32)).to(d),
                                    None if w is None else torch.from_numpy(np.asarray(w, np.float32)).to(d), n)
    ref = O.adjacency_from_edge_rows(src, dst, np.ones(len(src)) if w is None else w, n)
    assert np.array_equal(rp.cpu().numpy(), ref.indptr) and np.array_equal(col.cpu().numpy(), ref.indices)
    assert np.array_equal(val.cpu().numpy(), ref.data.astype(np.float32))


def test_edges_to_csr_reference_semantics():
    g = load_golden("weighted_small.npz")
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        for s in range(2):
            _ingest_case(g[p + "s%d_src" % s], g[p + "s%d_dst" % s], g[p + "s%d_w" % s], int(g[p + "n"]))
            # and against the reference's own matrix (utils.get_sp_adj_mat), bit for bit
            from ctgcn_amd import ops
            d = _dev()
            rp, col, val = ops.edges_to_csr(torch.from_numpy(g[p + "s%d_src" % s]).to(d), torch.from_numpy(g[p + "s%d_dst" % s]).to(d),
                                            torch.from_numpy(g[p + "s%d_w" % s].astype(np.float32)).to(d), int(g[p + "n"]))
            want = csr_from(g, p + "s%d_dateadj" % s, int(g[p + "n"]))
            assert np.array_equal(rp.cpu().numpy(), want.indptr) and np.array_equal(col.cpu().numpy(), want.indices)
            assert np.array_equal(val.cpu().numpy().astype(np.float64), want.data)
    snaps = load_golden("uci_snapshots.npz")          # 24 468 duplicate rows in 2004-05
    for t in range(7):
        _ingest_case(snaps["t%d_src" % t], snaps["t%d_dst" % t], snaps["t%d_w" % t], len(snaps["node_names"]))


def test_edges_to_csr_edge_cases_and_scale():
    rng = np.random.default_rng(3)
    _ingest_case(np.zeros(0, np.int32), np.zeros(0, np.int32), None, 5)                       # no rows
    _ingest_case([2, 2, 2], [2, 2, 2], [1.0, 2.0, 3.0], 4)                                    # only self loops
    _ingest_case([0, 1, 0, 1, 0], [1, 0, 1, 0, 1], [1.0, 2.0, 3.0, 4.0, 5.0], 2)              # one pair, five rows: last wins
    _ingest_case([3, 0], [0, 3], None, 4)                                                     # unweighted
    n, m = 200000, 3000000
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    src[:1000] = dst[:1000]                                                                    # self loops
    src[1000:200000], dst[1000:200000] = dst[400000:599000].copy(), src[400000:599000].copy()  # reversed duplicates
    _ingest_case(src, dst, rng.integers(1, 100, m).astype(np.float64), n)
    n = 46341                                                                                  # n*n just above 2^31
    src, dst = rng.integers(n - 50, n, 5000), rng.integers(n - 50, n, 5000)
    _ingest_case(src, dst, rng.integers(1, 9, 5000).astype(np.float64), n)


# ----------------------------------------------------------------------------------- hub rows
@pytest.mark.parametrize("d,long_row", [(128, 8), (500, 16), (12, 4), (128, 2048)])
def test_hub_rows_take_the_block_per_row_path(d, long_row):
    """rows longer than CoreAdj.LONG_ROW go through agg_*_hub_kernel; lowering the threshold pushes most rows there."""
    from ctgcn_amd import CoreAdj
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O
    n = 6000
    rng = np.random.default_rng(d)
    hub_nbrs = rng.choice(np.arange(1, n), 4500, replace=False)
    src = np.concatenate([np.zeros(4500, np.int64), rng.integers(1, n, 30000)])
    dst = np.concatenate([hub_nbrs, rng.integers(1, n, 30000)])
    csr = symmetric_csr_from_rows(src, dst, rng.integers(1, 5, len(src)) * 0.5, n)
    kept = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=6)[0]
    old = CoreAdj.LONG_ROW
    try:
        CoreAdj.LONG_ROW = long_row
        adj = _agg_case(kept, d, seed=long_row)
        assert adj.long_rows() is not None and adj.long_rows().numel() >= 1
        if long_row < 100:
            assert adj.long_rows().numel() > 100
    finally:
        CoreAdj.LONG_ROW = old


@pytest.mark.parametrize("d", [128, 500, 12])
def test_very_long_hub_rows_are_cut_into_pieces(d):
    """rows longer than 8192 entries: several blocks per row + a fixed-order second pass (forward, backward and the inference
    path's compact scratch), against the CPU oracle; two runs are bit-identical (nothing depends on block scheduling)."""
    from ctgcn_amd import CoreAdj, CoreDiffusion, ops
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O
    n = 60000
    rng = np.random.default_rng(d)
    src = np.concatenate([np.zeros(45000, np.int64), np.ones(20000, np.int64), rng.integers(2, n, 60000)])
    dst = np.concatenate([rng.choice(np.arange(2, n), 45000, replace=False), rng.choice(np.arange(2, n), 20000, replace=False),
                          rng.integers(2, n, 60000)])
    csr = symmetric_csr_from_rows(src, dst, rng.integers(1, 5, len(src)) * 0.5, n)
    kept = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=4)[0]
    adj = _agg_case(kept, d, seed=1, rtol=2e-5)          # forward + backward vs the oracle (45 000-term sums)
    assert adj.long_rows().numel() == 2 and adj.hub_split() == 6
    x = torch.randn(n, d, device=_dev(), requires_grad=True)
    runs = []
    for _ in range(2):
        x.grad = None
        H = ops.core_aggregate(x, adj)
        H.square().sum().backward()
        runs.append((H.detach().clone(), x.grad.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    if d % 4 == 0:                                          # inference: hub rows pass through the compact scratch of the split kernels
        torch.manual_seed(0)
        layer = CoreDiffusion(d, 128).to(_dev()).eval()
        with torch.no_grad():
            fused = layer(x.detach(), adj)
        seq = ops.core_aggregate(x.detach(), adj)
        from ctgcn_amd.layers import rnn_reduce_norm
        with torch.no_grad():
            plain = rnn_reduce_norm(layer.rnn, layer.norm, seq, reduce_sum=True)
        assert torch.isfinite(fused).all() and torch.equal(fused, plain)


def test_hub_rows_general_lists():
    from ctgcn_amd import CoreAdj
    rng = np.random.default_rng(1)
    n = 500
    mats = []
    for j in range(5):
        m = sp.random(n, n, density=0.05, random_state=j, format="lil", dtype=np.float32)
        m[3, :] = rng.standard_normal(n)            # a dense (asymmetric) hub row in every matrix
        mats.append(m.tocsr())
    old = CoreAdj.LONG_ROW
    try:
        CoreAdj.LONG_ROW = 64
        adj = _agg_case(mats, 40, seed=9, self_loop=False)
        assert not adj.nested and adj.long_rows().numel() >= 1
        assert adj.long_rows(transposed=True) is None or adj.long_rows(transposed=True).numel() >= 0
    finally:
        CoreAdj.LONG_ROW = old


@pytest.mark.parametrize("n,d,bias", [(1, 1, True), (31, 128, True), (1899, 128, True), (70001, 128, False), (4097, 50, True), (1000, 300, True)])
def test_linear_of_identity_is_transposed_weight_plus_bias(n, d, bias):
    """ctgcn_transpose_bias_f32: Linear applied to one-hot features (helper.py:161-172 identity through layers.py:95-106)
    equals W^T + b exactly (pure data movement + one add)."""
    from ctgcn_amd import ops
    lin = torch.nn.Linear(n, d, bias=bias).to(_dev())
    got = ops.linear_of_identity(lin.weight, lin.bias)
    want = lin.weight.detach().t() + (lin.bias.detach() if bias else 0.0)
    assert got.is_contiguous() and got.shape == (n, d)
    assert torch.equal(got, want.contiguous())


@pytest.mark.parametrize("n,d,bias", [(31, 128, True), (70001, 128, True), (4097, 50, False)])
def test_linear_of_identity_gradients(n, d, bias):
    """training: d weight = (d out)^T and d bias = column sums, through the same transpose kernel — exactly the framework's values"""
    from ctgcn_amd import ops
    torch.manual_seed(n)
    lin = torch.nn.Linear(n, d, bias=bias).to(_dev())
    g = torch.randn(n, d, device=_dev())
    out = ops.linear_of_identity(lin.weight, lin.bias)
    assert out.requires_grad and out.is_contiguous()
    (out * g).sum().backward()
    assert torch.equal(lin.weight.grad, g.t().contiguous())
    if bias:
        assert torch.allclose(lin.bias.grad, g.sum(0), rtol=1e-6, atol=1e-5)
