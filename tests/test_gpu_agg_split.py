"""ctgcn_core_aggregate_split_f32 (the aggregation writes fp16 planes + row scales instead of fp32 rows) feeding
ctgcn_linear_presplit_f32 (d_in != 128) or ctgcn_gru_layer_presplit_f32 (d_in = 128), against the separate kernels (fp32 H, then
ctgcn_linear_f32 / ctgcn_gru_layer_f32): bit-identical, including hub rows, padded widths and ragged tiles."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _layer(d, seed):
    from ctgcn_amd import CoreDiffusion
    torch.manual_seed(seed)
    return CoreDiffusion(d, 128).to(_dev()).eval()


def _nested_adj(n, m, max_core, seed, hub=0, with_mats=False):
    from ctgcn_amd import CoreAdj
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, n, m), rng.integers(0, n, m)
    if hub:
        src = np.concatenate([np.zeros(hub, np.int64), src])
        dst = np.concatenate([rng.choice(np.arange(1, n), hub, replace=False), dst])
    csr = symmetric_csr_from_rows(src, dst, rng.integers(1, 5, len(src)) * 0.5, n)
    kept = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=max_core)[0]
    adj = CoreAdj.from_matrices(kept, device=_dev())
    return (adj, kept) if with_mats else adj


def _deep_nested_adj(n, top, max_core, seed):
    """a nested list with more than 32 cores: a clique of `top` nodes, every other node attached to 1 .. top - 4 of them (its core number)"""
    from ctgcn_amd import CoreAdj
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    src, dst = [], []
    for i in range(top):
        for j in range(i):
            src.append(i); dst.append(j)
    for i in range(top, n):
        for j in rng.choice(top, 1 + (i * 7) % (top - 4), replace=False):
            src.append(i); dst.append(int(j))
    src, dst = np.array(src), np.array(dst)
    csr = symmetric_csr_from_rows(src, dst, rng.integers(1, 5, len(src)) * 0.5, n)
    kept = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=max_core)[0]
    return CoreAdj.from_matrices(kept, device=_dev())


def _both(layer, x, adj, monkeypatch, out_view=False):
    from ctgcn_amd import ops
    names = []
    ops.set_launch_timer(lambda name, s, e, meta: names.append((name, dict(meta))))
    try:
        with torch.no_grad():
            monkeypatch.setenv("CTGCN_AGG_SPLIT", "1")
            if out_view:
                buf = torch.zeros(x.shape[0], 3, 128, device=x.device)
                fused = layer(x, adj, out=buf[:, 1])
                assert fused.data_ptr() == buf[:, 1].data_ptr() and float(buf[:, 0].abs().max()) == 0 and float(buf[:, 2].abs().max()) == 0
            else:
                fused = layer(x, adj)
            used = [m for n_, m in names if n_ == "agg_fwd"]
            assert used and used[-1].get("split"), "the fused aggregation did not run"
            names.clear()
            monkeypatch.setenv("CTGCN_AGG_SPLIT", "0")
            plain = layer(x, adj)
            assert not any(m.get("split") for n_, m in names if n_ == "agg_fwd")
    finally:
        ops.set_launch_timer(None)
    torch.cuda.synchronize()
    return fused, plain


@pytest.mark.parametrize("d,n,m,max_core", [(500, 3000, 24000, 6), (64, 2000, 9000, 3), (256, 1500, 12000, 5), (260, 1200, 9000, 4),
                                            (36, 900, 5000, 2), (512, 700, 6000, 8), (500, 5, 4, 1),
                                            (128, 3001, 24000, 6), (128, 70001, 400000, 8), (128, 7, 9, 1), (128, 17, 40, 2)])
def test_fused_layer_is_bit_identical_to_the_separate_kernels(d, n, m, max_core, monkeypatch):
    adj = _nested_adj(n, m, max_core, seed=d + n)
    layer = _layer(d, d)
    x = torch.randn(n, d, device=_dev()) * torch.rand(n, 1, device=_dev()).exp()
    fused, plain = _both(layer, x, adj, monkeypatch)
    assert torch.isfinite(fused).all()
    assert torch.equal(fused, plain)


def test_fused_layer_with_hub_rows_and_strided_output(monkeypatch):
    from ctgcn_amd import CoreAdj
    old = CoreAdj.LONG_ROW
    try:
        CoreAdj.LONG_ROW = 12
        adj = _nested_adj(4000, 20000, 5, seed=3, hub=2500)
        assert adj.long_rows() is not None and adj.long_rows().numel() > 10
        for d in (500, 128):
            layer = _layer(d, 1)
            x = torch.randn(4000, d, device=_dev())
            fused, plain = _both(layer, x, adj, monkeypatch, out_view=True)
            assert torch.equal(fused, plain)
    finally:
        CoreAdj.LONG_ROW = old


@pytest.mark.parametrize("d", [500, 128, 96])
def test_fused_layer_on_general_matrix_lists(d, monkeypatch):
    """not nested, asymmetric, no unit diagonal (what a caller may pass instead of the loader's k-core list), with a dense hub row"""
    from ctgcn_amd import CoreAdj
    rng = np.random.default_rng(d)
    n, mats = 700, []
    for j in range(4):
        m = sp.random(n, n, density=0.02, random_state=10 * d + j, format="lil", dtype=np.float32)
        m[5, :] = rng.standard_normal(n)
        mats.append(m.tocsr())
    old = CoreAdj.LONG_ROW
    try:
        CoreAdj.LONG_ROW = 64
        adj = CoreAdj.from_matrices(mats, device=_dev(), self_loop=False)
        assert not adj.nested and adj.long_rows() is not None
        layer = _layer(d, 7)
        x = torch.randn(n, d, device=_dev())
        fused, plain = _both(layer, x, adj, monkeypatch)
        assert torch.isfinite(fused).all() and torch.equal(fused, plain)
    finally:
        CoreAdj.LONG_ROW = old


@pytest.mark.parametrize("d", [500, 128])
def test_fused_layer_matches_the_cpu_oracle(d):
    """CPU restatement of layers.py:41-62 on the same inputs; tolerance of the model parity tests (atol 1e-5 + rtol 1e-4)."""
    from oracle import torch_path as TP
    adj, kept = _nested_adj(800, 5000, 4, seed=11, with_mats=True)
    layer = _layer(d, 5)
    x = torch.randn(800, d, device=_dev())
    with torch.no_grad():
        got = layer(x, adj).cpu()
    sd = {k: v.detach().cpu() for k, v in layer.state_dict().items()}
    ref = TP.core_diffusion(sd, "", x.cpu(), [TP.coo_like_reference(m) for m in kept])
    assert torch.allclose(got, ref, atol=1e-5, rtol=1e-4), float((got - ref).abs().max())


def test_training_keeps_the_separate_path_and_the_switches_work(monkeypatch):
    from ctgcn_amd import ops
    monkeypatch.setenv("CTGCN_AGG_SPLIT", "1")
    adj = _nested_adj(500, 3000, 3, seed=2)
    layer = _layer(500, 2)
    x = torch.randn(500, 500, device=_dev(), requires_grad=True)
    assert not ops.aggregate_split_ok(layer.rnn, x, adj)
    layer(x, adj).sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    with torch.no_grad():
        assert ops.aggregate_split_ok(layer.rnn, x.detach(), adj)
        l128, x128 = _layer(128, 3), torch.randn(500, 128, device=_dev())
        assert ops.aggregate_split_ok(l128.rnn, x128, adj)
        monkeypatch.setenv("CTGCN_GRU_LAYER", "0")          # no layer kernel -> nothing consumes planes at width 128
        assert not ops.aggregate_split_ok(l128.rnn, x128, adj)
        monkeypatch.delenv("CTGCN_GRU_LAYER")
        monkeypatch.setenv("CTGCN_AGG_SPLIT", "0")
        assert not ops.aggregate_split_ok(l128.rnn, x128, adj) and not ops.aggregate_split_ok(layer.rnn, x.detach(), adj)


# ------------------------------------------------------------------ row plan: repeated rows of H are written / multiplied once
def _sparse_core_adj(n, seed, max_core=6):
    """mostly low-core nodes (many repeated leading rows), some isolated ones, a few dense ones"""
    from ctgcn_amd import CoreAdj
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    m = max(n // 2, 4)
    src, dst = rng.integers(0, n, m), (rng.integers(0, n, m) + 1 + np.arange(m) % 2) % n
    dense = rng.choice(n, min(n, max(8, n // 50)), replace=False)
    src = np.concatenate([src, rng.choice(dense, 12 * len(dense))])
    dst = np.concatenate([dst, rng.choice(dense, 12 * len(dense))])
    csr = symmetric_csr_from_rows(src, dst, np.ones(len(src)), n)
    kept = O.core_adj_list([O.kcore_matrices(csr)], 0, 1, 1, max_core=max_core)[0]
    return CoreAdj.from_matrices(kept, device=_dev()), kept


@pytest.mark.parametrize("n", [5000, 70001, 33, 16, 3])
def test_row_plan_skips_repeated_rows_and_changes_no_bit(n, monkeypatch):
    """CTGCN_DEDUP=1 (default) vs 0: same outputs bit for bit; the planes hold exactly the rows the tile masks name (holes untouched),
    and those rows are the plain path's rows at their planned positions."""
    from ctgcn_amd import ops
    adj, _ = _sparse_core_adj(n, seed=n)
    plan = adj.row_plan()
    K = adj.K
    assert plan is not None and plan["new_rows"] <= -(-n // 16) * 16 * K
    if n >= 5000:
        assert plan["new_rows"] < 0.8 * n * K, "this graph was built to have repeated rows"
    order = plan["order"].long()
    assert torch.equal(torch.sort(order).values, torch.arange(n, device=_dev()))
    layer = _layer(128, 1)
    x = torch.randn(n, 128, device=_dev())
    with torch.no_grad():
        monkeypatch.setenv("CTGCN_DEDUP", "1")
        with_plan = layer(x, adj)
        monkeypatch.setenv("CTGCN_DEDUP", "0")
        without = layer(x, adj)
    assert torch.isfinite(with_plan).all() and torch.equal(with_plan, without)
    # the planes themselves
    ws_plain, nbytes = ops.aggregate_split_planes(x, adj, 1)
    ws_plan = torch.full((nbytes,), 0x7b, dtype=torch.uint8, device=_dev())
    ops.aggregate_split_planes(x, adj, 1, plan, ws=ws_plan)
    torch.cuda.synchronize()
    rows = n * K
    def views(ws):
        p = ws[: rows * 128 * 4].view(torch.int16).view(2, rows, 128)
        s = ws[rows * 128 * 4: rows * 128 * 4 + rows * 4].view(torch.int32)
        return p, s
    (pa, sa), (pb, sb) = views(ws_plain), views(ws_plan)
    tm = plan["tile_mask"].long()
    pos = torch.arange(n, device=_dev())
    written = ((tm[pos // 16][:, None] >> torch.arange(K, device=_dev())[None, :]) & 1).bool()          # [position, slot]
    src = (order[:, None] * K + torch.arange(K, device=_dev())[None, :])                                # plain row of (position, slot)
    dst = (pos[:, None] * K + torch.arange(K, device=_dev())[None, :])
    assert torch.equal(pb[:, dst[written]], pa[:, src[written]]) and torch.equal(sb[dst[written]], sa[src[written]])
    hole = ~written
    assert bool((pb[:, dst[hole]] == 0x7b7b).all()) and bool((sb[dst[hole]] == 0x7b7b7b7b).all()), "a row the plan skips was written"
    # and a skipped row really is a repeat: the plain planes of slot j equal those of slot j - 1
    j_hole = torch.nonzero(hole)
    if j_hole.numel():
        p_, j_ = j_hole[:, 0], j_hole[:, 1]
        assert bool((j_ > 0).all())
        assert torch.equal(pa[:, order[p_] * K + j_], pa[:, order[p_] * K + j_ - 1])


def test_row_plan_with_hub_rows_general_lists_and_k1(monkeypatch):
    from ctgcn_amd import CoreAdj
    old = CoreAdj.LONG_ROW
    try:
        CoreAdj.LONG_ROW = 12
        adj, _ = _sparse_core_adj(6000, seed=5)
        assert adj.long_rows() is not None and adj.long_rows().numel() > 10
        cases = [adj]
        # not nested: slots without entries repeat the row before them anywhere in the list
        mats = [sp.random(900, 900, density=0.002 * (1 + j % 2), random_state=j, format="csr", dtype=np.float32) for j in range(5)]
        cases.append(CoreAdj.from_matrices(mats, device=_dev(), self_loop=False))
        cases.append(CoreAdj.from_matrices([mats[0]], device=_dev(), self_loop=True))
        many = CoreAdj.from_matrices([sp.random(300, 300, density=0.01, random_state=100 + j, format="csr", dtype=np.float32) for j in range(40)],
                                     device=_dev(), self_loop=False)
        # 33-64 slots (America-Air max core 64, Europe-Air 33, reference README.md:175-176): two mask words per tile
        assert many.K == 40 and many.row_plan() is not None and len(many.row_plan()["tile_mask"]) == 2 * -(-300 // 16)
        cases.append(many)
        cases.append(CoreAdj.from_matrices([sp.random(250, 250, density=0.0004 * (1 + 3 * (j % 5 == 0)), random_state=300 + j, format="csr", dtype=np.float32)
                                            for j in range(64)], device=_dev(), self_loop=False))
        assert cases[-1].K == 64 and cases[-1].row_plan() is not None and cases[-1].row_plan()["new_rows"] < -(-250 // 16) * 16 * 64      # repeats in both words
        # nested lists this deep: a dense graph whose cores reach 45 (max_core caps the list at 64 / 40)
        for top, mc in ((70, 64), (70, 40), (45, 64)):
            deep = _deep_nested_adj(600, top, mc, seed=17 + mc)
            assert 32 < deep.K <= mc and deep.row_plan() is not None and deep.row_plan()["new_rows"] < -(-600 // 16) * 16 * deep.K, deep.K
            cases.append(deep)
        too_long = CoreAdj.from_matrices([sp.random(100, 100, density=0.02, random_state=400 + j, format="csr", dtype=np.float32) for j in range(65)],
                                         device=_dev(), self_loop=False)
        assert too_long.K == 65 and too_long.row_plan() is None       # beyond 64 slots: no plan
        cases.append(too_long)
        full = CoreAdj.from_matrices([sp.random(300, 300, density=0.004, random_state=200 + j, format="csr", dtype=np.float32) for j in range(32)],
                                     device=_dev(), self_loop=False)
        assert full.K == 32 and full.row_plan() is not None and full.row_plan()["new_rows"] < 300 * 32      # bit 31 of the masks in use
        cases.append(full)
        for a in cases:
            for d in (128, 500, 64):          # 128: GRU layer kernel on planes with holes; others: split GEMM on compact operand rows
                layer = _layer(d, 2)
                x = torch.randn(a.n, d, device=_dev())
                with torch.no_grad():
                    monkeypatch.setenv("CTGCN_DEDUP", "1")
                    got = layer(x, a)
                    monkeypatch.setenv("CTGCN_DEDUP", "0")
                    want = layer(x, a)
                assert torch.isfinite(got).all() and torch.equal(got, want), (a.n, a.K, d)
                if a.K > 32:
                    # against the kernels that never see a step mask (fp32 H, then GEMM / layer kernel): round 4's plan-less layer kernel
                    # shifted its 32-bit all-ones mask by the step index — undefined from step 32 on, rows off by 1e-1 at K = 40 — and
                    # this test compared it with itself
                    with torch.no_grad():
                        monkeypatch.setenv("CTGCN_AGG_SPLIT", "0")
                        ref = layer(x, a)
                        monkeypatch.setenv("CTGCN_AGG_SPLIT", "1")
                    assert torch.equal(got, ref), (a.n, a.K, d)
    finally:
        CoreAdj.LONG_ROW = old


@pytest.mark.parametrize("d,n", [(500, 70001), (256, 5000), (96, 129), (500, 64), (500, 3)])
def test_compact_operand_rows_for_the_gemm_consumer(d, n, monkeypatch):
    """d != 128: under the row plan the aggregation writes only the rows that bring a new x, compactly; the split GEMM runs over those
    rows and the recurrence kernel reads a step's projection from the row of the last new x — the same numbers as without a plan."""
    from ctgcn_amd import ops
    adj, _ = _sparse_core_adj(n, seed=n + d)
    plan = adj.row_plan(adj.PLAN_TILE_GEMM)
    assert plan["operand_rows"] <= -(-n // 64) * 64 * adj.K
    if n >= 5000:
        assert plan["operand_rows"] < 0.8 * n * adj.K
    layer = _layer(d, 4)
    x = torch.randn(n, d, device=_dev())
    names = []
    ops.set_launch_timer(lambda name, s, e, meta: names.append((name, dict(meta))))
    try:
        with torch.no_grad():
            monkeypatch.setenv("CTGCN_DEDUP", "1")
            got = layer(x, adj)
            rows_gemm = [m["rows"] for nm, m in names if nm == "linear_split"]
            monkeypatch.setenv("CTGCN_DEDUP", "0")
            want = layer(x, adj)
    finally:
        ops.set_launch_timer(None)
    assert rows_gemm == [plan["operand_rows"]]
    assert torch.isfinite(got).all() and torch.equal(got, want)
