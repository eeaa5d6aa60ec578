"""Training form of a CoreDiffusion layer with d_in = hidden = 128 (ops._CoreDiffusionFused: planes + row plan forward,
ctgcn_gru_layer_presplit_save_f32 -> ctgcn_layernorm_bwd_f32 -> ctgcn_gru_bwd_rec_f32 -> ctgcn_gru_bwd_in_f32 -> ctgcn_core_aggregate_bwd_f32
backward) against float64 autograd of the reference's CPU path (oracle/torch_path.py, layers.py:38-63) and against round 3's path.

Tolerance for gradients (as tests/test_gpu_models.py): within 1e-4 of the tensor's largest entry, checked against FLOAT64 truth — the kernels'
bf16 x 2 products carry 2^-17 per operand, an fp32 CPU run of the same expression is itself ~1e-6 away from it."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _graph(n, avg_deg, seed, max_core):
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import dynamic_graph
    from oracle import oracle as O, torch_path as TP
    g = dynamic_graph(n, avg_deg=avg_deg, snapshots=1, seed=seed)[0]
    adj, _, _ = core_adj_from_scipy(g, max_core, torch.device(DEV))
    ref = [TP.coo_like_reference(m) for m in O.core_adj_list([O.kcore_matrices(g)], 0, 1, 1, max_core)[0]]
    return adj, ref


def _layer(seed):
    from ctgcn_amd.layers import CoreDiffusion
    torch.manual_seed(seed)
    layer = CoreDiffusion(128, 128)
    with torch.no_grad():
        layer.norm.weight.uniform_(0.5, 1.5)
        layer.norm.bias.uniform_(-0.5, 0.5)
    return layer


def _truth(layer, x, ref_adj, G):
    """float64 autograd of layers.py:38-63 on the CPU"""
    from oracle import torch_path as TP
    sd = {"l." + k: v.detach().double().clone().requires_grad_(True) for k, v in layer.state_dict().items() if not k.startswith("linear.")}
    xd = x.detach().double().clone().requires_grad_(True)
    adj64 = [a.double() for a in ref_adj]
    saved = TP._rnn
    TP._rnn = TP._rnn_grad
    try:
        out = TP.core_diffusion(sd, "l.", xd, adj64)
    finally:
        TP._rnn = saved
    (out * G.double()).sum().backward()
    grads = {k[2:]: v.grad for k, v in sd.items()}
    grads["x"] = xd.grad
    return out.detach(), grads


def _run(layer_cpu, x, adj, G, fused):
    import copy
    layer = copy.deepcopy(layer_cpu).to(DEV)
    xg = x.to(DEV).clone().requires_grad_(True)
    old = os.environ.get("CTGCN_TRAIN_FUSED")
    os.environ["CTGCN_TRAIN_FUSED"] = "1" if fused else "0"
    try:
        out = layer(xg, adj)
        (out * G.to(DEV)).sum().backward()
    finally:
        if old is None:
            os.environ.pop("CTGCN_TRAIN_FUSED", None)
        else:
            os.environ["CTGCN_TRAIN_FUSED"] = old
    grads = {k: p.grad.detach().cpu() for k, p in layer.named_parameters() if p.grad is not None}
    grads["x"] = xg.grad.detach().cpu()
    return out.detach().cpu(), grads


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("n,avg_deg,max_core,seed", [(50, 4, 3, 1), (777, 8, 6, 2), (4100, 12, 8, 3), (3000, 6, -1, 4)])
def test_fused_training_layer_matches_float64_autograd(n, avg_deg, max_core, seed):
    from ctgcn_amd import ops
    adj, ref_adj = _graph(n, avg_deg, seed, max_core)
    layer = _layer(seed)
    torch.manual_seed(100 + seed)
    x = torch.randn(n, 128)
    G = torch.randn(n, 128)
    assert ops.core_diffusion_fused_ok(layer.rnn.to(DEV), layer.norm.to(DEV), x.to(DEV), adj)
    layer = layer.cpu()
    want_out, want = _truth(layer, x, ref_adj, G)
    got_out, got = _run(layer, x, adj, G, fused=True)
    assert (got_out.double() - want_out).abs().max().item() < 5e-5
    for k in ("x", "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0", "norm.weight", "norm.bias"):
        assert k in got, k
        assert _rel(got[k], want[k]) < 1e-4, (k, _rel(got[k], want[k]))


def test_fused_training_layer_against_round3_path_and_determinism():
    """same layer, same inputs through round 3's kernels (fp32 H, bf16 x 3 backward): both within tolerance of each other; two fused
    runs bit-identical (per-block partial sums, fixed reduction order — no atomics anywhere)"""
    adj, _ = _graph(5000, 10, 7, 8)
    layer = _layer(7)
    torch.manual_seed(8)
    x, G = torch.randn(5000, 128), torch.randn(5000, 128)
    out_a, ga = _run(layer, x, adj, G, fused=True)
    out_b, gb = _run(layer, x, adj, G, fused=False)
    out_c, gc = _run(layer, x, adj, G, fused=True)
    assert (out_a - out_b).abs().max().item() < 2e-5
    for k in ga:
        assert _rel(ga[k], gb[k]) < 1e-4, (k, _rel(ga[k], gb[k]))
        assert torch.equal(ga[k], gc[k]), k
    assert torch.equal(out_a, out_c)


def test_fused_training_layer_in_row_chunks():
    """the backward walks the positions in chunks (bounded gates buffer): a chunked run equals the one-chunk run bit for bit"""
    from ctgcn_amd import ops
    adj, _ = _graph(20000, 8, 9, 8)
    layer = _layer(9)
    torch.manual_seed(10)
    x, G = torch.randn(20000, 128), torch.randn(20000, 128)
    _, g1 = _run(layer, x, adj, G, fused=True)
    old = ops._GI_MAX_ELEMS
    try:
        ops._GI_MAX_ELEMS = 1          # -> chunks of one row granule
        _, g2 = _run(layer, x, adj, G, fused=True)
    finally:
        ops._GI_MAX_ELEMS = old
    for k in g1:
        if k == "x":
            assert torch.equal(g1[k], g2[k]), k        # per-row results do not depend on the chunking
        else:
            assert _rel(g1[k], g2[k]) < 1e-5, k        # weight gradients: another association of the per-block partial sums


def test_fused_training_without_row_plan(monkeypatch):
    """CTGCN_DEDUP=0: no plan — every step fresh, natural row order, same kernels"""
    monkeypatch.setenv("CTGCN_DEDUP", "0")
    adj, ref_adj = _graph(1000, 8, 11, 5)
    layer = _layer(11)
    torch.manual_seed(12)
    x, G = torch.randn(1000, 128), torch.randn(1000, 128)
    want_out, want = _truth(layer, x, ref_adj, G)
    got_out, got = _run(layer, x, adj, G, fused=True)
    assert (got_out.double() - want_out).abs().max().item() < 5e-5
    for k in want:
        assert _rel(got[k], want[k]) < 1e-4, (k, _rel(got[k], want[k]))


@pytest.mark.parametrize("z_bias", [6.0, 12.0, 30.0])
def test_fused_training_with_saturated_update_gate(z_bias):
    """The recompute pass keeps r, z, q and the backward rebuilds n = h_{t-1} + (h_t - h_{t-1}) / (1 - z): update gates close to (and, in
    fp32, equal to) 1 must not hurt — every use of n carries the factor (1 - z) it was divided by."""
    adj, ref_adj = _graph(1200, 8, 21, 6)
    layer = _layer(21)
    with torch.no_grad():
        layer.rnn.bias_ih_l0[128:256] += z_bias          # z = sigmoid(... + z_bias): 0.9975 / 1 - 6e-6 / exactly 1.0f
    torch.manual_seed(22)
    x, G = torch.randn(1200, 128), torch.randn(1200, 128)
    want_out, want = _truth(layer, x, ref_adj, G)
    got_out, got = _run(layer, x, adj, G, fused=True)
    assert torch.isfinite(got_out).all() and all(torch.isfinite(v).all() for v in got.values())
    # reference point: round 3's path (all four gates stored) on the same inputs — the LayerNorm behind a nearly frozen state (h moves by
    # (1 - z) per step) amplifies every path's rounding, so the bound is "not worse than the path that stores n"
    _, old = _run(layer, x, adj, G, fused=False)
    report = {}
    for k in want:
        scale = want[k].abs().max().item()
        err = (got[k].double() - want[k]).abs().max().item()
        err_old = (old[k].double() - want[k]).abs().max().item()
        report[k] = (err, err_old, scale)
    for k, (e_new, e_old, scale) in report.items():
        # relative to the tensor's largest entry as everywhere else, with an absolute floor: at z = 1.0f every gradient through the GRU is
        # ~1e-10 (the state never moves) and "relative to the largest entry" compares rounding noise with rounding noise
        assert e_new <= max(1e-4 * scale, 3.0 * e_old) + 1e-7, (k, e_new, e_old, scale)


def test_planes_are_written_again_beyond_the_memory_budget():
    """ops._CoreDiffusionFused keeps a layer's operand planes for its backward only while the kept bytes stay under the budget
    (CTGCN_TRAIN_PLANES_GB); beyond it the backward writes them again from the layer's input: the same kernel on the same input — every
    gradient bit-identical — and the accounting returns to zero when the graph is gone (also for a forward that never ran its backward)."""
    import gc
    from ctgcn_amd import ops
    adj, _ = _graph(6000, 10, 31, 8)
    layer = _layer(31)
    torch.manual_seed(32)
    x, G = torch.randn(6000, 128), torch.randn(6000, 128)
    assert ops._kept_planes["bytes"] == 0
    _, kept = _run(layer, x, adj, G, fused=True)
    old = ops._kept_planes["budget"]
    try:
        ops._kept_planes["budget"] = 0
        _, again = _run(layer, x, adj, G, fused=True)
    finally:
        ops._kept_planes["budget"] = old
    for k in kept:
        assert torch.equal(kept[k], again[k]), k
    dev_layer = layer.to(DEV)
    out = dev_layer(x.to(DEV).requires_grad_(True), adj)       # a forward whose backward never runs
    assert ops._kept_planes["bytes"] > 0
    del out
    gc.collect()
    assert ops._kept_planes["bytes"] == 0


@pytest.mark.parametrize("top,mc", [(50, 40), (70, 64)])
@pytest.mark.parametrize("dedup", ["1", "0"])
def test_fused_training_layer_with_core_lists_of_33_to_64_matrices(top, mc, dedup, monkeypatch):
    """Round 6 (VERDICT r5 missing 4): core lists of 33-64 matrices (America-Air max core 64, Europe-Air 33: reference README.md:175-176) train
    through the fused path too — the row plan's two mask words per tile in the recompute pass (gru_layer8_h2_kernel<planes, sum, SAVE, WIDE>),
    gru_bwd_rec_kernel<., WIDE> and gru_bwd_in_kernel's 64-bit masks; without a plan the masks are all ones over 64 bits.  Graph: a clique of
    `top` nodes + nodes attached to parts of it (k-cores up to top - 1, kept: the `mc` deepest).  Gradients against float64 autograd of the
    reference's path; rows of a clique saturate the gates over 40-64 steps, so the fp32 CPU path itself is the yardstick for the tolerance
    where 1e-4 of the largest entry is too tight (as tests/test_gpu_models.py::test_core_lists_deeper_than_32...)."""
    from ctgcn_amd import CoreAdj, ops
    from ctgcn_amd.utils import symmetric_csr_from_rows
    from oracle import oracle as O, torch_path as TP
    monkeypatch.setenv("CTGCN_DEDUP", dedup)
    n = 400
    rng = np.random.default_rng(31)
    src, dst = [], []
    for i in range(top):
        for j in range(i):
            src.append(i); dst.append(j)
    for i in range(top, n):
        for j in rng.choice(top, 1 + (i * 7) % (top - 4), replace=False):
            src.append(i); dst.append(int(j))
    g = symmetric_csr_from_rows(np.array(src), np.array(dst), rng.integers(1, 5, len(src)) * 0.25, n)
    kept = O.core_adj_list([O.kcore_matrices(g)], 0, 1, 1, mc)[0]
    assert 32 < len(kept) <= mc
    adj = CoreAdj.from_matrices(kept, device=DEV)
    ref_adj = [TP.coo_like_reference(m) for m in kept]
    layer = _layer(9)
    torch.manual_seed(109)
    x = torch.randn(n, 128) * 0.1
    G = torch.randn(n, 128)
    assert ops.core_diffusion_fused_ok(layer.rnn.to(DEV), layer.norm.to(DEV), x.to(DEV), adj)
    assert (adj.row_plan() is not None) == (dedup == "1") or dedup == "0"
    layer = layer.cpu()
    want_out, want = _truth(layer, x, ref_adj, G)
    got_out, got = _run(layer, x, adj, G, fused=True)
    old_out, old = _run(layer, x, adj, G, fused=False)          # round 3's path (fp32 H, no plan): what these lists trained on before
    assert (got_out.double() - want_out).abs().max().item() < 2e-4
    for k in ("x", "rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0", "norm.weight", "norm.bias"):
        assert k in got, k
        new_err, old_err = _rel(got[k], want[k]), _rel(old[k], want[k])
        assert new_err < max(1e-4, 2.0 * old_err), (k, new_err, old_err)
