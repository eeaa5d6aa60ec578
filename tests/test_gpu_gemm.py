"""ctgcn_linear_f32 (fp16x2 split GEMM on the matrix cores) against float64: the GRU input projection at d_in = 500 and the
dense nn.Linear layers of the MLP run through it.  Bar: never less accurate than the fp32 library GEMM it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rows,k,n_out,bias", [
    (1, 32, 1, True), (127, 36, 5, False), (1000, 500, 384, True), (4097, 500, 128, True), (300, 1740, 500, True),
    (129, 128, 129, False), (60_730, 500, 500, True), (5000, 1737, 500, True), (333, 61, 7, False),
])
def test_split_gemm_is_fp32_accurate(rows, k, n_out, bias):
    from ctgcn_amd import ops
    torch.manual_seed(rows + k)
    x = torch.randn(rows, k, device=DEV) * torch.rand(rows, 1, device=DEV).mul(6).exp()       # row magnitudes over e^6
    w = torch.randn(n_out, k, device=DEV) / k ** 0.5
    b = torch.randn(n_out, device=DEV) if bias else None
    assert ops.linear_split_ok(x, w)
    got = ops.linear_split(x, w, b)
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    lib32 = torch.addmm(b, x, w.t()) if bias else x @ w.t()
    scale = (x.double().abs() @ w.double().abs().t()) + 1e-30                   # sum |x_k w_k|: the natural error scale of a dot product
    err_split = ((got.double() - ref).abs() / scale).max().item()
    err_lib = ((lib32.double() - ref).abs() / scale).max().item()
    print("rows=%d k=%d n=%d: max err / sum|xw|  split %.2e  fp32 library %.2e" % (rows, k, n_out, err_split, err_lib))
    assert err_split <= max(2 * err_lib, 2e-7), (err_split, err_lib)


def test_split_gemm_strided_views_and_special_rows():
    from ctgcn_amd import ops
    torch.manual_seed(3)
    big = torch.randn(500, 3, 512, device=DEV)
    x = big[:, 1, :500]                                   # row stride 1536, 500 columns
    x[7] = 0.0                                            # all-zero row
    x[8] *= 1e30
    x[9] *= 1e-30
    w = torch.randn(384, 512, device=DEV)[:, :500]
    out = torch.full((500, 400), 7.0, device=DEV)
    assert ops.linear_split_ok(x, w)
    ops.linear_split(x, w, None, out=out[:, :384])
    ref = x.double() @ w.double().t()
    scale = (x.double().abs() @ w.double().abs().t()) + 1e-300
    assert torch.isfinite(out).all() and bool((out[:, 384:] == 7.0).all()) and bool((out[7, :384] == 0).all())
    assert ((out[:, :384].double() - ref).abs() / scale).max().item() <= 4e-7


def test_gru_with_wide_input_uses_the_split_gemm_and_matches_torch():
    """nn.GRU(500 -> 128): the shape of the first CoreDiffusion layer of every shipped config (hid_dim 500)."""
    from ctgcn_amd import ops
    torch.manual_seed(5)
    rnn = torch.nn.GRU(500, 128, 1, batch_first=True)
    norm = torch.nn.LayerNorm(128)
    x = torch.relu(torch.randn(3000, 5, 500))
    with torch.no_grad():
        want = norm(rnn(x)[0].sum(1))
        seen = []
        ops.set_launch_timer(lambda name, s, e, meta: seen.append(name))
        got = ops.gru_sequence(rnn.to(DEV), x.to(DEV), norm.to(DEV), True)
        ops.set_launch_timer(None)
    assert "linear_split" in seen and "gru_seq" in seen
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("rows,k,n_out", [(5000, 1737, 500), (777, 503, 130), (64, 33, 9)])
def test_rows_that_are_only_4_byte_aligned_and_selu_epilogue(rows, k, n_out):
    """k % 4 != 0: rows start at 4-byte aligned addresses and are read with 16-byte loads + a scalar tail — the result equals the GEMM
    on a zero-padded, 16-byte aligned copy bit for bit (same planes).  activation = SELU in the epilogue equals F.selu of the plain
    output (torch's selu uses expm1 as the kernel does) to a few ulp, and is the reference's Linear + SELU (layers.py:95-106)."""
    from ctgcn_amd import ops
    torch.manual_seed(k)
    x = torch.randn(rows, k, device=DEV) * 3
    w = torch.randn(n_out, k, device=DEV) / k ** 0.5
    b = torch.randn(n_out, device=DEV)
    kpad = -(-k // 4) * 4
    xp = torch.zeros(rows, kpad + 4, device=DEV)[:, :kpad]          # 16-byte aligned rows, zero columns beyond k
    wp = torch.zeros(n_out, kpad + 4, device=DEV)[:, :kpad]
    xp[:, :k] = x
    wp[:, :k] = w
    assert x.stride(0) % 4 != 0 or k % 4 == 0
    got = ops.linear_split(x, w, b)
    assert torch.equal(got, ops.linear_split(xp, wp, b))
    act = ops.linear_split(x, w, b, selu=True)
    want = torch.nn.functional.selu(got)
    assert torch.allclose(act, want, rtol=2e-6, atol=1e-7), float((act - want).abs().max())
    assert bool((act[got > 0] > 0).all()) and bool((act[got < 0] < 0).all())


def test_operand_plane_cache_follows_the_tensors(monkeypatch):
    """ops._PlaneCache: planes of the weights / static inputs are split once, an in-place update re-splits them, results equal the
    uncached call bit for bit"""
    import torch
    from ctgcn_amd import ops
    dev = "cuda:0"
    torch.manual_seed(3)
    x = torch.randn(777, 1737, device=dev)
    w = torch.nn.Parameter(torch.randn(500, 1737, device=dev) * 0.05)
    b = torch.randn(500, device=dev)
    monkeypatch.setenv("CTGCN_GEMM", "hand")                 # bit identity cached / uncached: the hand-written GEMM's planes
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "0")
    ref = ops.linear_split(x, w, b, selu=True)
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "1")
    ops._plane_cache.entries.clear(); ops._plane_cache.bytes = 0
    a1 = ops.linear_split(x, w, b, selu=True, static_x=True)
    assert len(ops._plane_cache.entries) == 2
    a2 = ops.linear_split(x, w, b, selu=True, static_x=True)
    assert torch.equal(a1, ref) and torch.equal(a2, ref)
    with torch.no_grad():
        w.mul_(2.0)                      # what optimizer.step() does: in place, version counter bumps
        x[0, 0] = 5.0
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "0")
    ref2 = ops.linear_split(x, w, b, selu=True)
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "1")
    a3 = ops.linear_split(x, w, b, selu=True, static_x=True)
    assert torch.equal(a3, ref2) and not torch.equal(a3, ref)
    del x, w
    import gc
    gc.collect()
    assert len(ops._plane_cache.entries) == 0        # entries die with their tensors


def test_mlp_chain_matches_float64(monkeypatch):
    """ops.mlp_chain_split: three Linear + SELU layers with operand planes handed from epilogue to main loop (per-(row, 128-column) scales,
    accumulators rescaled at the k-block boundaries) against float64, and against the layer-by-layer path"""
    import torch
    from ctgcn_amd.layers import MLP
    dev = "cuda:0"
    torch.manual_seed(5)
    for rows, widths in ((1000, (1737, 500, 500, 128)), (333, (96, 200, 64)), (70, (64, 130, 500, 32))):
        mlp = MLP(widths[0], widths[1], widths[-1], len(widths) - 1, activate_type='N')
        # give the hidden widths of the test (MLP builds in -> hid ... hid -> out)
        lins = [torch.nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])]
        mlp.linears = torch.nn.ModuleList(lins)
        mlp.layer_num = len(lins)
        x = torch.randn(rows, widths[0]) * torch.rand(rows, 1) * 10
        want = x.double()
        for lin in lins:
            want = torch.nn.functional.selu(torch.nn.functional.linear(want, lin.weight.double(), lin.bias.double()))
        mlp = mlp.to(dev).eval()
        xg = x.to(dev)
        monkeypatch.setenv("CTGCN_GEMM", "hand")        # the chain belongs to the hand-written GEMM
        with torch.no_grad():
            monkeypatch.setenv("CTGCN_MLP_CHAIN", "1")
            got = mlp(xg)
            monkeypatch.setenv("CTGCN_MLP_CHAIN", "0")
            ref = mlp(xg)
        scale = want.abs().max().item()
        e_chain = (got.double().cpu() - want).abs().max().item() / scale
        e_layer = (ref.double().cpu() - want).abs().max().item() / scale
        assert e_chain < 2e-6, (widths, e_chain, e_layer)
        assert e_chain < 4 * e_layer + 1e-7, (widths, e_chain, e_layer)


@pytest.mark.parametrize("rows,k,n_out", [(1000, 500, 384), (60_730, 500, 500), (5000, 1737, 500), (300, 1740, 128), (129, 128, 132)])
def test_library_gemm_over_k3_planes_against_the_hand_kernel(rows, k, n_out, monkeypatch):
    """The default dense path (ops.mlp_k3: ONE library fp16 GEMM over [hi | lo | hi] x [lo | hi | hi]^T, scales / bias in the finishing kernel)
    and the hand-written split GEMM (CTGCN_GEMM=hand) against float64: same error class (fp16 x 2 operands, fp32 accumulation)."""
    from ctgcn_amd import ops
    torch.manual_seed(rows + k + n_out)
    x = torch.randn(rows, k, device=DEV) * torch.rand(rows, 1, device=DEV).mul(6).exp()
    w = torch.randn(n_out, k, device=DEV) / k ** 0.5
    w[3] *= 1e-3                                          # a small row under the tensor-wide weight scale
    b = torch.randn(n_out, device=DEV)
    ref = x.double() @ w.double().t() + b.double()
    scale = (x.double().abs() @ w.double().abs().t()) + 1e-30
    seen = []
    ops.set_launch_timer(lambda name, s, e, meta: seen.append(meta.get("library", False)) if name == "linear_split" else None)
    try:
        monkeypatch.setenv("CTGCN_GEMM", "lib")
        got = ops.linear_split(x, w, b)
        assert seen and all(seen), seen
        del seen[:]
        monkeypatch.setenv("CTGCN_GEMM", "hand")
        hand = ops.linear_split(x, w, b)
        assert seen and not any(seen), seen
    finally:
        ops.set_launch_timer(None)
    e_lib = ((got.double() - ref).abs() / scale).max().item()
    e_hand = ((hand.double() - ref).abs() / scale).max().item()
    lib32 = torch.addmm(b, x, w.t())
    e_32 = ((lib32.double() - ref).abs() / scale).max().item()
    print("rows=%d k=%d n=%d: library-k3 %.2e  hand %.2e  fp32 library %.2e" % (rows, k, n_out, e_lib, e_hand, e_32))
    assert e_lib <= max(2 * e_32, 2e-7), (e_lib, e_hand, e_32)


def test_mlp_k3_chain_matches_float64(monkeypatch):
    """three Linear + SELU layers (layers.py:95-106) on the library path: a layer's scales, bias and SELU are applied by the next layer's
    split — against float64 and against the hand-written layer-by-layer path"""
    from ctgcn_amd.layers import MLP
    torch.manual_seed(15)
    for rows, widths in ((1000, (1737, 500, 500, 128)), (333, (96, 200, 64)), (70, (64, 132, 500, 32))):
        lins = [torch.nn.Linear(a, b) for a, b in zip(widths[:-1], widths[1:])]
        mlp = MLP(widths[0], widths[1], widths[-1], len(widths) - 1, activate_type='N')
        mlp.linears = torch.nn.ModuleList(lins)
        mlp.layer_num = len(lins)
        x = torch.randn(rows, widths[0]) * torch.rand(rows, 1) * 10
        want = x.double()
        for lin in lins:
            want = torch.nn.functional.selu(torch.nn.functional.linear(want, lin.weight.double(), lin.bias.double()))
        mlp = mlp.to(DEV).eval()
        xg = x.to(DEV)
        seen = []
        from ctgcn_amd import ops
        ops.set_launch_timer(lambda name, s, e, meta: seen.append(meta.get("library", False)) if name == "linear_split" else None)
        try:
            with torch.no_grad():
                monkeypatch.setenv("CTGCN_GEMM", "lib")
                got = mlp(xg)
                assert len(seen) == len(lins) and all(seen), seen
                monkeypatch.setenv("CTGCN_GEMM", "hand")
                ref = mlp(xg)
        finally:
            ops.set_launch_timer(None)
        scale = want.abs().max().item()
        e_lib = (got.double().cpu() - want).abs().max().item() / scale
        e_hand = (ref.double().cpu() - want).abs().max().item() / scale
        assert e_lib < 2e-6, (widths, e_lib, e_hand)
        assert e_lib < 4 * e_hand + 1e-7, (widths, e_lib, e_hand)


def test_k3_planes_are_cached_and_follow_in_place_updates(monkeypatch):
    """the library path keeps the k3 planes of weights / static inputs (ops._PlaneCache.planes_k3): same result from the cache, a new one
    after an in-place update (version counter), entries die with their tensors"""
    from ctgcn_amd import ops
    monkeypatch.setenv("CTGCN_GEMM", "lib")
    torch.manual_seed(4)
    x = torch.randn(600, 500, device=DEV)
    w = torch.nn.Parameter(torch.randn(384, 500, device=DEV) * 0.05)
    b = torch.randn(384, device=DEV)
    ops._plane_cache.entries.clear(); ops._plane_cache.bytes = 0
    a1 = ops.linear_split(x, w, b, static_x=True)
    assert len(ops._plane_cache.entries) == 2
    a2 = ops.linear_split(x, w, b, static_x=True)
    assert torch.equal(a1, a2)
    with torch.no_grad():
        w.mul_(2.0)
        x[0, 0] = 5.0
    a3 = ops.linear_split(x, w, b, static_x=True)
    ref = x.double() @ w.detach().double().t() + b.double()          # detached: an autograd graph would keep w alive
    assert ((a3.double() - ref).abs().max() / ref.abs().max()).item() < 1e-6 and not torch.equal(a3, a1)
    del x, w
    import gc
    gc.collect()
    assert len(ops._plane_cache.entries) == 0


def test_direct_to_lds_operand_staging_is_bit_identical(monkeypatch):
    """CTGCN_GEMM_DMA=1 (global_load_lds_dwordx4 instead of registers + ds_write_b128): the same tiles, the same MFMA order — the same bits"""
    from ctgcn_amd import ops
    monkeypatch.setenv("CTGCN_GEMM", "hand")
    torch.manual_seed(11)
    for rows, k, n_out in ((5000, 500, 384), (777, 1737, 500), (129, 96, 130)):
        x = torch.randn(rows, k, device=DEV) * torch.rand(rows, 1, device=DEV).mul(4).exp()
        w = torch.randn(n_out, k, device=DEV) / k ** 0.5
        b = torch.randn(n_out, device=DEV)
        monkeypatch.setenv("CTGCN_GEMM_DMA", "0")
        ref = ops.linear_split(x, w, b, selu=True)
        monkeypatch.setenv("CTGCN_GEMM_DMA", "1")
        got = ops.linear_split(x, w, b, selu=True)
        assert torch.equal(got, ref), (rows, k, n_out)
        monkeypatch.setenv("CTGCN_GEMM_DMA", "0")
        monkeypatch.setenv("CTGCN_GEMM_WIDE", "1")            # 256 x 128 tiles, eight waves, three LDS stages
        got = ops.linear_split(x, w, b, selu=True)
        monkeypatch.setenv("CTGCN_GEMM_WIDE", "0")
        assert torch.equal(got, ref), ("wide", rows, k, n_out)
