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
    """ops._PlaneCache: the packed operand of a weight and the planes of a static input are built once, an in-place update re-builds them,
    results equal the uncached call bit for bit; writes through .data need ops.invalidate_plane_cache() (documented limit of the version
    counter); tensors made under torch.inference_mode() are never cached"""
    import gc
    from ctgcn_amd import ops
    torch.manual_seed(3)
    x = torch.randn(777, 1737, device=DEV)
    w = torch.nn.Parameter(torch.randn(500, 1737, device=DEV) * 0.05)
    b = torch.randn(500, device=DEV)
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "0")
    ref = ops.linear_split(x, w, b, selu=True)
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "1")
    ops.invalidate_plane_cache()
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
    w.data.mul_(0.5)                     # behind the version counter's back: stale until invalidated
    assert torch.equal(ops.linear_split(x, w, b, selu=True, static_x=True), ref2)
    ops.invalidate_plane_cache()
    assert len(ops._plane_cache.entries) == 0
    a4 = ops.linear_split(x, w, b, selu=True, static_x=True)
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "0")
    assert torch.equal(a4, ops.linear_split(x, w, b, selu=True))
    monkeypatch.setenv("CTGCN_PLANE_CACHE", "1")
    del x, w
    gc.collect()
    assert len(ops._plane_cache.entries) == 0        # entries die with their tensors
    with torch.inference_mode():                     # no version counter: split per call, nothing cached, no exception
        xi = torch.randn(300, 96, device=DEV)
        wi = torch.randn(64, 96, device=DEV)
        y1 = ops.linear_split(xi, wi, None, static_x=True)
        y2 = ops.linear_split(xi, wi, None, static_x=True)
    assert len(ops._plane_cache.entries) == 0 and torch.equal(y1, y2)
    assert ((y1.double() - xi.double() @ wi.double().t()).abs().max() / (xi.double().abs() @ wi.double().abs().t()).max()).item() < 1e-6


def test_model_runs_under_inference_mode():
    """a CTGCN-S built and run inside torch.inference_mode() (dense features through the MLP's split GEMMs, W_ih through the projection)"""
    import ctgcn_amd
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import dynamic_graph
    graphs = dynamic_graph(1500, avg_deg=8, snapshots=2, seed=3)
    adj = [core_adj_from_scipy(g, 4, DEV)[0] for g in graphs]
    torch.manual_seed(1)
    xs = [torch.randn(1500, 200) for _ in graphs]
    model = ctgcn_amd.CTGCN(200, 96, 128, 2, 1, 2, model_type="S", trans_activate_type="N").to(DEV).eval()
    with torch.no_grad():
        want, _ = model([x.to(DEV) for x in xs], adj)
    with torch.inference_mode():
        got, _ = model([x.to(DEV) for x in xs], adj)
    assert torch.equal(got, want)


def test_static_features_are_opt_in():
    """only a tensor marked with ops.mark_static (what helper.DataLoader returns) gets its planes cached by the MLP"""
    from ctgcn_amd import ops
    from ctgcn_amd.layers import MLP
    torch.manual_seed(2)
    mlp = MLP(300, 64, 32, 2, activate_type='N').to(DEV).eval()
    x = torch.randn(500, 300, device=DEV)
    ops.invalidate_plane_cache()
    with torch.no_grad():
        y0 = mlp(x)
        assert all(e[0]() is not x for e in ops._plane_cache.entries.values())
        x.data.add_(1.0)                       # an unmarked input may be edited any way the caller likes
        y1 = mlp(x)
        assert not torch.equal(y0, y1)
        ops.mark_static(x)
        y2 = mlp(x)
        assert torch.equal(y1, y2) and any(e[0]() is x for e in ops._plane_cache.entries.values())


@pytest.mark.parametrize("rows,k,n_out", [(128, 64, 16), (129, 64, 500), (255, 100, 384), (256, 1737, 512), (3000, 500, 513), (700, 96, 1100), (40_000, 500, 384)])
def test_panel_kernel_edges(rows, k, n_out):
    """gemm_h2_panel_kernel at its seams: one k stage (k <= 64: the scales of a panel are staged in the same stage its epilogue runs),
    partial last panels, every column-tile count per wave (N = 16 ... 512), more than 512 columns (one launch per chunk), more panels
    than CUs (persistent blocks walk several panels: the cross-panel prefetch), SELU on / off, output written into a wider buffer."""
    from ctgcn_amd import ops
    torch.manual_seed(rows + n_out)
    x = torch.randn(rows, k, device=DEV) * torch.rand(rows, 1, device=DEV).mul(4).exp()
    w = torch.randn(n_out, k, device=DEV) / k ** 0.5
    b = torch.randn(n_out, device=DEV)
    ref = x.double() @ w.double().t() + b.double()
    scale = (x.double().abs() @ w.double().abs().t()) + 1e-30
    buf = torch.full((rows, n_out + 5), 7.0, device=DEV)             # ldy % 4 != 0 for some shapes: the scalar store path
    got = ops.linear_split(x, w, b, out=buf[:, :n_out])
    assert bool((buf[:, n_out:] == 7.0).all())
    assert ((got.double() - ref).abs() / scale).max().item() <= 4e-7
    act = ops.linear_split(x, w, b, selu=True)
    assert torch.allclose(act, torch.nn.functional.selu(got), rtol=2e-6, atol=1e-7)
    again = ops.linear_split(x, w, b, out=torch.empty(rows, n_out, device=DEV))
    assert torch.equal(again, got)                                    # aligned (16-byte stores) and unaligned outputs: the same values


@pytest.mark.parametrize("rows,k,n_out", [(1500, 200, 96), (40_000, 500, 128), (40_000, 500, 500), (300_000, 500, 384)])
def test_panel_kernel_is_deterministic_under_repetition(rows, k, n_out):
    """Twenty launches on the same operands, bit for bit — the check that caught the round-5 variant with inline-asm W-fragment loads (wrong
    128-row panels in one run of ten at n_out <= 128, invisible to a single accuracy test): few panels per block, one k stage per panel,
    several panels per block, all CUs busy."""
    from ctgcn_amd import ops
    torch.manual_seed(rows)
    x = torch.randn(rows, k, device=DEV)
    w = torch.randn(n_out, k, device=DEV) / k ** 0.5
    b = torch.randn(n_out, device=DEV)
    ref = ops.linear_split(x, w, b)
    scale = (x.double().abs() @ w.double().abs().t()) + 1e-30
    assert ((ref.double() - (x.double() @ w.double().t() + b.double())).abs() / scale).max().item() <= 4e-7
    for _ in range(20):
        assert torch.equal(ops.linear_split(x, w, b), ref)


@pytest.mark.parametrize("rows,k,n_out,bias", [(5000, 1737, 500, True), (4100, 500, 128, True), (6000, 96, 40, False)])
def test_linear_under_autograd_matches_float64(rows, k, n_out, bias):
    """ops._LinearSplit (MLP._linear in training): y and dx on the split GEMM, dW / db from the library — against float64 autograd of
    nn.Linear (reference layers.py:95-106), errors no larger than the fp32 module's own"""
    from ctgcn_amd import ops
    from ctgcn_amd.layers import MLP
    torch.manual_seed(rows + k)
    lin = torch.nn.Linear(k, n_out, bias=bias).to(DEV)
    x = torch.randn(rows, k, device=DEV, requires_grad=True)
    G = torch.randn(rows, n_out, device=DEV)
    lin64 = torch.nn.Linear(k, n_out, bias=bias).to(DEV).double()
    lin64.load_state_dict({n: p.detach().double() for n, p in lin.named_parameters()})
    x64 = x.detach().double().requires_grad_(True)
    y64 = lin64(x64)
    (y64 * G.double()).sum().backward()
    want = [y64.detach(), x64.grad, lin64.weight.grad] + ([lin64.bias.grad] if bias else [])

    def run(split):
        for p in list(lin.parameters()) + [x]:
            p.grad = None
        if split:
            y, fused = MLP._linear(lin, x, False)
            assert not fused and type(y.grad_fn).__name__ == "_LinearSplitBackward"
        else:
            y = lin(x)
        (y * G).sum().backward()
        return [y.detach(), x.grad.clone(), lin.weight.grad.clone()] + ([lin.bias.grad.clone()] if bias else [])

    got, plain = run(True), run(False)
    for name, g, p, w in zip(("y", "dx", "dW", "db"), got, plain, want):
        scale = float(w.abs().max())
        e_split, e_fp32 = float((g.double() - w).abs().max()) / scale, float((p.double() - w).abs().max()) / scale
        print("  [tol] linear autograd %-3s |err| / max: split path %.3e, fp32 module %.3e" % (name, e_split, e_fp32))
        assert e_split <= max(2.0 * e_fp32, 2e-6), (name, e_split, e_fp32)


def test_mlp_training_with_split_linears_against_float64(monkeypatch):
    """the MLP of CTGCN-S (3 Linear layers, SELU after each: reference layers.py:95-106) under autograd with its Linear layers on
    ops._LinearSplit, and on torch's own (CTGCN_LINEAR_TRAIN=0), both against float64 autograd of the same module: the split path's errors
    stay within a small factor of the library path's.  (A whole CTGCN-S step cannot be compared entry by entry between the two: they round
    `trans` differently, a pre-activation of relu(cumulative A x) at rounding distance of zero flips, and because a weight gradient sums over
    all rows every entry of it moves by ~1e-3 of the scale — measured, with the CoreDiffusion gradients agreeing to 1e-5.)"""
    import copy
    from ctgcn_amd.layers import MLP
    torch.manual_seed(9)
    rows = 6000
    mlp = MLP(300, 500, 128, 3, activate_type="N").to(DEV).train()
    x = torch.randn(rows, 300, device=DEV)
    G = torch.randn(rows, 128, device=DEV)
    mlp64 = copy.deepcopy(mlp).double()
    y64 = mlp64(x.double())
    (y64 * G.double()).sum().backward()
    want = [y64.detach()] + [p.grad for p in mlp64.parameters()]
    errs = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("CTGCN_LINEAR_TRAIN", flag)
        for p in mlp.parameters():
            p.grad = None
        y = mlp(x)
        (y * G).sum().backward()
        got = [y.detach()] + [p.grad for p in mlp.parameters()]
        errs[flag] = [float((g.double() - w).abs().max()) / float(w.abs().max()) for g, w in zip(got, want)]
    names = ["y"] + [n for n, _ in mlp.named_parameters()]
    for n_, a, b in zip(names, errs["1"], errs["0"]):
        print("  [tol] MLP autograd %-18s |err| / max vs float64: split linears %.3e, library linears %.3e" % (n_, a, b))
        assert a <= max(3.0 * b, 5e-6), (n_, a, b)


@pytest.mark.parametrize("rows,k,n_out,selu,bias", [(6000, 1737, 500, True, True), (6000, 500, 500, True, True), (129, 96, 40, False, True),
                                                     (70001, 200, 384, False, False), (300, 64, 512, True, True), (1, 40, 33, True, True)])
def test_chained_output_planes_equal_the_split_of_the_fp32_output(rows, k, n_out, selu, bias):
    """ctgcn_linear_packed_chain_f32: the GEMM's output as the next layer's operand planes — bit for bit what ctgcn_split_rows_f32 makes of the
    fp32 output of ctgcn_linear_packed_f32 (same epilogue arithmetic, row maximum over all eight waves' columns, zero padding to 64 columns)"""
    from ctgcn_amd import ops, _lib
    lib = _lib.load()
    torch.manual_seed(rows + k + n_out)
    x = torch.randn(rows, k, device=DEV)
    w = torch.randn(n_out, k, device=DEV) / k ** 0.5
    b = torch.randn(n_out, device=DEV) if bias else None
    y = ops.linear_split(x, w, b, selu=selu)
    nbytes = int(lib.ctgcn_split_planes_bytes(rows, n_out))
    want = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    _lib.check(lib.ctgcn_split_rows_f32(rows, n_out, y.data_ptr(), y.stride(0), want.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream), "split")
    got = ops.linear_split(x, w, b, selu=selu, planes_out=True)
    assert isinstance(got, ops.Planes) and got.shape == (rows, n_out) and got.buf.numel() == nbytes
    kp = -(-n_out // 64) * 64
    used = rows * kp * 4 + rows * 4
    torch.cuda.synchronize()
    assert torch.equal(got.buf[:used], want[:used])
    # and the consumer takes them: the same fp32 rows as from the tensor itself
    w2 = torch.randn(77, n_out, device=DEV)
    if n_out >= 32:
        assert torch.equal(ops.linear_split(got, w2, None), ops.linear_split(y, w2, None))


def test_mlp_inference_chain_is_bit_identical(monkeypatch):
    """MLP.forward in inference: hidden activations go from GEMM to GEMM as operand planes (CTGCN_MLP_CHAIN, default on) — the bits of the
    layer-by-layer path (fp32 rows written, split again), for the CTGCN-S transform shape (reference layers.py:95-106) and a linear one"""
    from ctgcn_amd import ops
    from ctgcn_amd.layers import MLP
    for dims, act, rows in (((1737, 500, 128, 3), "N", 6000), ((300, 64, 40, 4), "L", 1000), ((96, 500, 128, 2), "N", 129)):
        torch.manual_seed(dims[0])
        mlp = MLP(dims[0], dims[1], dims[2], dims[3], activate_type=act).to(DEV).eval()
        x = torch.randn(rows, dims[0], device=DEV)
        calls = []
        real = ops.linear_split
        monkeypatch.setattr(ops, "linear_split", lambda *a_, **k_: (calls.append(bool(k_.get("planes_out"))), real(*a_, **k_))[1])
        with torch.no_grad():
            monkeypatch.setenv("CTGCN_MLP_CHAIN", "1")
            got = mlp(x)
            assert calls == [True] * (dims[3] - 1) + [False], calls
            monkeypatch.setenv("CTGCN_MLP_CHAIN", "0")
            want = mlp(x)
        monkeypatch.setattr(ops, "linear_split", real)
        assert torch.isfinite(got).all() and torch.equal(got, want), dims
