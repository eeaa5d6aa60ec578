"""Fused GRU recurrent kernel (ctgcn_gru_seq_f32) vs stock torch.nn.GRU (+ sum / LayerNorm) in fp32 on the CPU."""
import numpy as np
import pytest
import torch

from conftest import check_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(rnn, norm, x, reduce_sum):
    with torch.no_grad():
        out = rnn(x)[0]
        out = out.sum(1) if reduce_sum else out
        return norm(out) if norm is not None else out


@pytest.mark.parametrize("rows,steps,d_in,reduce_sum,bias,use_norm", [
    (1, 1, 128, True, True, True), (33, 8, 128, True, True, True), (1000, 8, 500, True, True, True),
    (4097, 5, 128, False, True, True), (300, 16, 128, False, True, True), (64, 3, 24, True, False, True),
    (257, 22, 128, True, True, False), (100, 2, 128, False, False, False),
])
def test_fused_gru_matches_torch(rows, steps, d_in, reduce_sum, bias, use_norm, split_mode):
    from ctgcn_amd import ops
    torch.manual_seed(rows + steps)
    rnn = torch.nn.GRU(d_in, 128, 1, bias=bias, batch_first=True)
    norm = torch.nn.LayerNorm(128) if use_norm else None
    if norm is not None:
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5)
            norm.bias.uniform_(-0.5, 0.5)
    x = torch.relu(torch.randn(rows, steps, d_in)) * 2.0
    want = _ref(rnn, norm, x, reduce_sum)
    rnn_d = rnn.to(DEV)
    norm_d = norm.to(DEV) if norm is not None else None
    with torch.no_grad():
        assert ops.gru_fused_ok(rnn_d, x.to(DEV))
        got = ops.gru_sequence(rnn_d, x.to(DEV), norm_d, reduce_sum)
    check_close(got.cpu().numpy(), want.numpy(), 1e-4, 1e-5, what="fused GRU vs torch")


def test_fused_gru_row_chunking_and_determinism():
    from ctgcn_amd import ops
    torch.manual_seed(0)
    rnn = torch.nn.GRU(128, 128, 1, batch_first=True).to(DEV)
    norm = torch.nn.LayerNorm(128).to(DEV)
    x = torch.randn(5000, 8, 128, device=DEV)
    with torch.no_grad():
        a = ops.gru_sequence(rnn, x, norm, True)
        old = ops._GI_MAX_ELEMS
        try:
            ops._GI_MAX_ELEMS = 1000 * 8 * 384          # force 5 chunks (+ ragged tail)
            b = ops.gru_sequence(rnn, x, norm, True)
        finally:
            ops._GI_MAX_ELEMS = old
        c = ops.gru_sequence(rnn, x, norm, True)
    assert torch.equal(a, c)
    assert (a - b).abs().max().item() < 1e-5        # hipBLASLt may pick another GEMM kernel for another M


@pytest.mark.parametrize("rows,steps,d_in,reduce_sum,bias,use_norm", [
    (70, 8, 128, True, True, True), (1000, 5, 40, True, True, True), (513, 6, 128, False, True, True),
    (90, 3, 128, True, False, True), (33, 1, 16, True, True, True), (200, 4, 128, False, True, False),
    (700, 5, 500, True, True, True), (130, 3, 200, True, True, True),      # d_in > 128: dW_ih by column slices (ops.wide_weight_grad_enabled)
])
def test_fused_gru_gradients_match_torch_autograd(rows, steps, d_in, reduce_sum, bias, use_norm):
    """d/d{x, W_ih, W_hh, b_ih, b_hh, ln.weight, ln.bias} of sum(out * G) vs CPU nn.GRU autograd."""
    import copy
    from ctgcn_amd import ops
    torch.manual_seed(7 * rows + steps)
    rnn = torch.nn.GRU(d_in, 128, 1, bias=bias, batch_first=True)
    norm = torch.nn.LayerNorm(128) if use_norm else None
    if norm is not None:
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5)
            norm.bias.uniform_(-0.5, 0.5)
    x = (torch.relu(torch.randn(rows, steps, d_in)) * 1.5).requires_grad_(True)
    out = rnn(x)[0]
    out = out.sum(1) if reduce_sum else out
    out = norm(out) if norm is not None else out
    G = torch.randn_like(out)
    (out * G).sum().backward()

    rnn_d, norm_d = copy.deepcopy(rnn).to(DEV), (copy.deepcopy(norm).to(DEV) if norm is not None else None)
    for p in list(rnn_d.parameters()) + (list(norm_d.parameters()) if norm_d is not None else []):
        p.grad = None
    xd = x.detach().to(DEV).requires_grad_(True)
    got = ops.gru_sequence(rnn_d, xd, norm_d, reduce_sum)
    check_close(got.detach().cpu().numpy(), out.detach().numpy(), 1e-4, 1e-5, what="fused GRU (training forward) vs torch")
    (got * G.to(DEV)).sum().backward()

    def close(a, b, name):
        a, b = a.cpu().numpy(), b.numpy()
        scale = max(1e-6, float(np.abs(b).max()))
        print("  [tol] grad %-24s |err| / max|grad| %.3e (limit 1e-5)" % (name, np.abs(a - b).max() / scale))
        assert np.abs(a - b).max() <= 1e-5 * scale, (name, np.abs(a - b).max(), scale)      # observed worst 5.5e-7

    close(xd.grad, x.grad, "dx")
    for (name, pd), (_, pc) in zip(rnn_d.named_parameters(), rnn.named_parameters()):
        close(pd.grad, pc.grad, name)
    if norm is not None:
        close(norm_d.weight.grad, norm.weight.grad, "ln.weight")
        close(norm_d.bias.grad, norm.bias.grad, "ln.bias")


def test_training_and_inference_paths_agree():
    """with gradients enabled the same fused forward runs (plus a HIP backward); outputs are identical."""
    import ctgcn_amd
    torch.manual_seed(1)
    layer = ctgcn_amd.CoreDiffusion(128, 128).to(DEV)
    import scipy.sparse as sp
    m = sp.random(500, 500, 0.02, random_state=1, format="csr", dtype=np.float32)
    m = (m + m.T).tocsr()
    adj = ctgcn_amd.CoreAdj.from_matrices([m + sp.eye(500, format="csr", dtype=np.float32)], device=DEV)
    x = torch.randn(500, 128, device=DEV, requires_grad=True)
    y_train = layer(x, adj)
    assert y_train.requires_grad
    y_train.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    with torch.no_grad():
        y_inf = layer(x, adj)
    assert torch.equal(y_inf, y_train.detach())


@pytest.fixture(params=["f16x2", "bf16x3"])
def split_mode(request, monkeypatch):
    """both 16-bit split arithmetics of the forward kernels (default fp16x2; CTGCN_GRU_SPLIT=bf16x3)"""
    from ctgcn_amd import ops
    monkeypatch.setenv("CTGCN_GRU_SPLIT", request.param)
    monkeypatch.setenv("CTGCN_FP32_MFMA_ONLY", "0")
    assert ops.forward_split_mode() == (2 if request.param == "f16x2" else 1)
    return request.param


@pytest.mark.parametrize("rows", [1, 63, 64, 1000, 70000])
def test_split_bf16_projection_is_fp32_accurate(rows, split_mode):
    """ctgcn_gru_input_proj_f32 (fp16x2: per-row scaled two-term fp16 split, three products; bf16x3: three-term bf16 split,
    six products) vs an fp64 reference: its error must not exceed the error of a plain fp32 GEMM of the same operands by
    more than a small factor, and both are ~1e-6."""
    from ctgcn_amd import ops
    torch.manual_seed(rows)
    x = (torch.relu(torch.randn(rows, 128)) * 4.0)
    w = (torch.rand(384, 128) - 0.5) * 0.18
    b = torch.randn(384) * 0.1
    ref = x.double() @ w.double().t() + b.double()
    out = torch.empty(rows, 384, device=DEV)
    assert ops.split_mfma_enabled()
    ops._project(x.to(DEV), w.to(DEV), b.to(DEV), out)
    err_split = (out.cpu().double() - ref).abs().max().item()
    err_fp32 = ((x @ w.t() + b).double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err_split <= max(2.0 * err_fp32, 2e-7 * scale), (err_split, err_fp32, scale)
    out2 = torch.empty(rows, 384, device=DEV)
    ops._project(x.to(DEV), w.to(DEV), None, out2)
    assert torch.allclose(out2.cpu() + b, out.cpu(), atol=1e-5)


@pytest.mark.parametrize("rows,steps,d_in,reduce_sum,bias", [
    (1, 1, 128, True, True), (100, 8, 128, True, True), (1000, 5, 300, False, True), (4100, 12, 64, True, False),
])
def test_fused_lstm_matches_torch(rows, steps, d_in, reduce_sum, bias):
    from ctgcn_amd import layers, ops
    torch.manual_seed(rows * 3 + steps)
    rnn = torch.nn.LSTM(d_in, 128, 1, bias=bias, batch_first=True)
    norm = torch.nn.LayerNorm(128)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x = torch.relu(torch.randn(rows, steps, d_in)) * 2.0
    want = _ref(rnn, norm, x, reduce_sum)
    rnn_d, norm_d = rnn.to(DEV), norm.to(DEV)
    with torch.no_grad():
        assert ops.lstm_fused_ok(rnn_d, x.to(DEV))
        got = layers.rnn_reduce_norm(rnn_d, norm_d, x.to(DEV), reduce_sum)
    check_close(got.cpu().numpy(), want.numpy(), 1e-4, 1e-5, what="fused GRU vs torch")
    # with gradients enabled the same kernels run behind an autograd function (and agree)
    got_train = layers.rnn_reduce_norm(rnn_d, norm_d, x.to(DEV).requires_grad_(True), reduce_sum)
    assert got_train.requires_grad
    check_close(got_train.detach().cpu().numpy(), want.numpy(), 1e-4, 1e-5, what="fused LSTM (training path) vs torch")


def test_split_bf16_projection_wide_dynamic_range(split_mode):
    """operands spanning 10 orders of magnitude inside a row (and exact zeros / negatives), all-zero rows, rows of huge
    (1e30) and of tiny (1e-30) entries, weight rows of very different magnitude: the split keeps fp32 accuracy."""
    from ctgcn_amd import ops
    torch.manual_seed(5)
    rows = 4096
    mag = 10.0 ** torch.empty(rows, 128).uniform_(-6, 4)
    x = mag * torch.sign(torch.randn(rows, 128)) * (torch.rand(rows, 128) > 0.1)
    x[7] = 0.0
    x[8] *= 1e26
    x[9] *= 1e-20
    x[10, 1:] = 0.0
    w = (10.0 ** torch.empty(384, 128).uniform_(-4, 0)) * torch.sign(torch.randn(384, 128))
    w[5] *= 1e-12
    w[6] *= 1e6
    w[300] = 0.0
    ref = x.double() @ w.double().t()
    out = torch.empty(rows, 384, device=DEV)
    ops._project(x.to(DEV), w.to(DEV), None, out)
    # error measured against the magnitude of the terms that were summed (cancellation-aware bound)
    scale = (x.abs().double() @ w.abs().double().t())
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    zero = scale == 0                                   # all-zero row of x or of w: the result is exactly 0
    assert bool((got[zero] == 0).all()) and int(zero.sum()) >= 128 + 4096
    scale = torch.where(zero, torch.ones_like(scale), scale)
    err_split = ((got - ref).abs() / scale).max().item()
    err_fp32 = (((x @ w.t()).double() - ref).abs() / scale).max().item()
    assert err_split <= max(2.0 * err_fp32, 3e-7), (err_split, err_fp32)


@pytest.mark.parametrize("rows", [1, 31, 32, 1000, 70001])
def test_split_bf16_input_gradient_is_fp32_accurate(rows):
    """ctgcn_gru_input_grad_f32: dX = dGI · W_ih against fp64; error no worse than a plain fp32 GEMM's."""
    from ctgcn_amd import ops
    torch.manual_seed(rows + 7)
    g = torch.randn(rows, 384) * torch.rand(rows, 1) * 3.0
    w = (torch.rand(384, 128) - 0.5) * 0.18
    ref = g.double() @ w.double()
    out = torch.full((rows, 128), float("nan"), device=DEV)
    assert ops.split_mfma_enabled()
    ops._project_grad(g.to(DEV), w.to(DEV), out)
    err_split = (out.cpu().double() - ref).abs().max().item()
    err_fp32 = ((g @ w).double() - ref).abs().max().item()
    assert err_split <= max(2.0 * err_fp32, 2e-7 * ref.abs().max().item()), (err_split, err_fp32)


@pytest.mark.parametrize("nodes,steps,shift", [(1, 1, False), (1, 1, True), (5, 3, True), (37, 8, False), (37, 8, True),
                                               (4099, 8, True), (9000, 16, True), (20000, 6, False), (1000, 5, True), (1000, 5, False),
                                               (333, 7, True), (6000, 3, True)])
def test_split_bf16_weight_gradient_is_fp32_accurate(nodes, steps, shift):
    """ctgcn_gru_weight_grad_f32: dW = sum_r [g01[:, :256] | g2]^T x' (x' optionally shifted one step inside each sequence,
    i.e. h_{t-1} from the h sequence) against fp64; error no worse than a plain fp32 GEMM's.  Row counts that are not
    multiples of the 32-row chunk, fewer chunks than block pairs, strided g01."""
    from ctgcn_amd import ops
    torch.manual_seed(nodes * 17 + steps)
    R = nodes * steps
    dgi = torch.randn(R, 384) * 0.5
    dghn = torch.randn(R, 128) * 0.5
    x = torch.tanh(torch.randn(nodes, steps, 128))
    if shift:
        xs = torch.zeros_like(x)
        xs[:, 1:] = x[:, :-1]
        G = torch.cat([dgi[:, :256], dghn], 1)
    else:
        xs = x
        G = dgi
    ref = G.double().t() @ xs.reshape(R, 128).double()
    part = torch.full((ops._DW_PAIRS, 384, 128), float("nan"), device=DEV)
    dgi_d = dgi.to(DEV)
    ops._weight_grad(part, dgi_d, dghn.to(DEV) if shift else dgi_d[:, 256:], x.reshape(R, 128).to(DEV), steps, shift, False)
    out = part.sum(0)
    err_split = (out.cpu().double() - ref).abs().max().item()
    err_fp32 = ((G.t() @ xs.reshape(R, 128)).double() - ref).abs().max().item()
    assert err_split <= max(2.0 * err_fp32, 3e-7 * ref.abs().max().item()), (err_split, err_fp32)
    ops._weight_grad(part, dgi_d, dghn.to(DEV) if shift else dgi_d[:, 256:], x.reshape(R, 128).to(DEV), steps, shift, True)
    assert torch.allclose(part.sum(0).cpu().double(), 2 * ref, rtol=1e-5, atol=1e-5 * ref.abs().max().item())   # accumulate flag


@pytest.mark.parametrize("rows,steps", [(1, 1), (63, 2), (64, 8), (1000, 5), (70001, 8), (40000, 16), (600_000, 7)])
@pytest.mark.parametrize("reduce_sum", [True, False])
def test_register_resident_layer_kernel_equals_the_kernel_pair(rows, steps, reduce_sum, monkeypatch):
    """ctgcn_gru_layer_f32 (projection + recurrence in one kernel, both weight matrices in the register file, gi consumed from
    the accumulators) performs the kernel pair's arithmetic operation for operation: bit-identical outputs, with and without
    bias / LayerNorm, dense and strided outputs."""
    from ctgcn_amd import ops
    torch.manual_seed(rows + steps)
    # (600 000 rows = 146 tiles per block: the first 8-wave build staged the summed rows in an x slot other waves were still
    # reading — invisible at 70 001 rows, non-finite outputs at 1 M)
    for bias, use_norm in ((True, True), (False, False)) if rows < 500_000 else ((True, True),):
        rnn = torch.nn.GRU(128, 128, 1, bias=bias, batch_first=True).to(DEV)
        norm = torch.nn.LayerNorm(128).to(DEV) if use_norm else None
        if norm is not None:
            with torch.no_grad():
                norm.weight.uniform_(0.5, 1.5)
                norm.bias.uniform_(-0.5, 0.5)
        x = torch.relu(torch.randn(rows, steps, 128, device=DEV)) * torch.rand(rows, 1, 1, device=DEV) * 3
        with torch.no_grad():
            monkeypatch.setenv("CTGCN_GRU_LAYER", "0")
            want = ops.gru_sequence(rnn, x, norm, reduce_sum)
            monkeypatch.setenv("CTGCN_GRU_LAYER", "all")
            assert ops.layer_kernel_enabled(reduce_sum)
            got = ops.gru_sequence(rnn, x, norm, reduce_sum)
            assert torch.isfinite(got).all()
            assert torch.equal(got, want), float((got - want).abs().max())
            if reduce_sum:                                   # strided output view (column t of a [rows, T, 128] tensor)
                buf = torch.zeros(rows, 3, 128, device=DEV)
                ops.gru_sequence(rnn, x, norm, True, out=buf[:, 1])
                assert torch.equal(buf[:, 1], want) and not buf[:, 0].any() and not buf[:, 2].any()


@pytest.mark.parametrize("rows,steps", [(1, 1), (1003, 1), (777, 5), (70001, 8)])
def test_layernorm_backward_kernel_matches_autograd(rows, steps):
    """ctgcn_layernorm_bwd_f32: d/dx and d/d(gamma, beta) of LayerNorm(sum_t h[:, t]) — against float64 autograd of the same expression."""
    from ctgcn_amd import _lib
    from ctgcn_amd._lib import check, ptr
    lib = _lib.load()
    torch.manual_seed(rows + steps)
    dev = torch.device("cuda:0")
    h = torch.randn(rows, steps, 128, device=dev) * 0.7
    dy = torch.randn(rows, 128, device=dev)
    gamma = torch.randn(128, device=dev)
    dx = torch.empty(rows, 128, device=dev)
    part = torch.empty(300, 256, device=dev)
    big = torch.randn(rows, 3, 128, device=dev)            # dy as a column of a [rows, 3, 128] gradient: read in place through ld_dy
    dy = big[:, 1]
    check(lib.ctgcn_layernorm_bwd_f32(rows, steps, 128, ptr(h), ptr(dy), dy.stride(0), ptr(gamma), 1e-5, ptr(dx), ptr(part), part.shape[0], None,
                                      torch.cuda.current_stream().cuda_stream), "ctgcn_layernorm_bwd_f32")
    x64 = h.double().sum(1).requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), torch.zeros(128, dtype=torch.float64, device=dev, requires_grad=True)
    torch.nn.functional.layer_norm(x64, (128,), g64, b64, 1e-5).backward(dy.double())
    sums = part.double().sum(0)
    assert (dx.double() - x64.grad).abs().max().item() <= 2e-5 * x64.grad.abs().max().item() + 1e-6
    assert (sums[:128] - g64.grad).abs().max().item() <= 1e-5 * g64.grad.abs().max().item() + 1e-5
    assert (sums[128:] - b64.grad).abs().max().item() <= 1e-5 * b64.grad.abs().max().item() + 1e-5


@pytest.mark.parametrize("rows,steps,d_in,reduce_sum,bias,use_norm", [
    (70, 8, 128, True, True, True), (1000, 5, 40, True, True, True), (513, 6, 128, False, True, True),
    (90, 3, 128, True, False, True), (33, 1, 16, True, True, True), (200, 4, 128, False, True, False), (70001, 3, 128, True, True, True),
])
def test_fused_lstm_gradients_match_torch_autograd(rows, steps, d_in, reduce_sum, bias, use_norm):
    """rnn_type = 'LSTM' in training (reference layers.py:27-28, models.py:234-235): d/d{x, W_ih, W_hh, b_ih, b_hh, ln.weight, ln.bias}
    of sum(out * G) through ctgcn_lstm_seq_f32 / ctgcn_lstm_seq_bwd_f32 / ctgcn_layernorm_bwd_f32 vs CPU nn.LSTM autograd."""
    import copy
    from ctgcn_amd import ops
    torch.manual_seed(11 * rows + steps)
    rnn = torch.nn.LSTM(d_in, 128, 1, bias=bias, batch_first=True)
    norm = torch.nn.LayerNorm(128) if use_norm else None
    if norm is not None:
        with torch.no_grad():
            norm.weight.uniform_(0.5, 1.5)
            norm.bias.uniform_(-0.5, 0.5)
    x = (torch.relu(torch.randn(rows, steps, d_in)) * 1.5).requires_grad_(True)
    out = rnn(x)[0]
    out = out.sum(1) if reduce_sum else out
    out = norm(out) if norm is not None else out
    G = torch.randn_like(out)
    (out * G).sum().backward()

    rnn_d, norm_d = copy.deepcopy(rnn).to(DEV), (copy.deepcopy(norm).to(DEV) if norm is not None else None)
    for p in list(rnn_d.parameters()) + (list(norm_d.parameters()) if norm_d is not None else []):
        p.grad = None
    xd = x.detach().to(DEV).requires_grad_(True)
    assert ops.lstm_fused_ok(rnn_d, xd)
    got = ops.lstm_sequence(rnn_d, xd, norm_d, reduce_sum)
    check_close(got.detach().cpu().numpy(), out.detach().numpy(), 1e-4, 1e-5, what="fused LSTM (training forward) vs torch")
    (got * G.to(DEV)).sum().backward()

    def close(a, b, name):
        a, b = a.cpu().numpy(), b.numpy()
        scale = max(1e-6, float(np.abs(b).max()))
        print("  [tol] grad %-24s |err| / max|grad| %.3e (limit 2e-5)" % (name, np.abs(a - b).max() / scale))
        assert np.abs(a - b).max() <= 2e-5 * scale, (name, np.abs(a - b).max(), scale)

    close(xd.grad, x.grad, "dx")
    for (name, pd), (_, pc) in zip(rnn_d.named_parameters(), rnn.named_parameters()):
        close(pd.grad, pc.grad, name)
    if norm is not None:
        close(norm_d.weight.grad, norm.weight.grad, "ln.weight")
        close(norm_d.bias.grad, norm.bias.grad, "ln.bias")


def test_kept_projection_gives_the_recomputed_gradients(monkeypatch):
    """d_in = 500: the backward starts its recompute pass from the forward's gi (kept under the training buffer budget) or projects x again
    (CTGCN_KEEP_GI=0, or a second backward through the same graph): the same bits either way, and the budget's accounting returns to zero"""
    import gc
    from ctgcn_amd import ops
    torch.manual_seed(11)
    rnn = torch.nn.GRU(500, 128, 1, batch_first=True).to(DEV)
    norm = torch.nn.LayerNorm(128).to(DEV)
    x = (torch.relu(torch.randn(900, 6, 500, device=DEV)) * 1.5).requires_grad_(True)
    G = torch.randn(900, 128, device=DEV)
    base = ops._kept_planes["bytes"]

    def grads(keep, twice=False):
        monkeypatch.setenv("CTGCN_KEEP_GI", keep)
        for p in list(rnn.parameters()) + list(norm.parameters()) + [x]:
            p.grad = None
        out = ops.gru_sequence(rnn, x, norm, True)
        if keep == "1":
            assert ops._kept_planes["bytes"] > base
        (out * G).sum().backward(retain_graph=twice)
        first = [p.grad.clone() for p in list(rnn.parameters()) + list(norm.parameters()) + [x]]
        if twice:                                           # the kept buffer was consumed: this one recomputes; gradients accumulate to 2 x
            (out * G).sum().backward()
            second = [p.grad.clone() for p in list(rnn.parameters()) + list(norm.parameters()) + [x]]
            for a, b in zip(first, second):
                assert torch.allclose(b, 2 * a, rtol=1e-6, atol=1e-7)
        return [out.detach().clone()] + first

    a = grads("1")
    b = grads("0")
    c = grads("1", twice=True)
    for u, v, w in zip(a, b, c):
        assert torch.equal(u, v) and torch.equal(u, w)
    del a, b, c
    gc.collect()
    torch.cuda.synchronize()
    assert ops._kept_planes["bytes"] == base
