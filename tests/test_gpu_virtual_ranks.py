"""The snapshot-parallel CTGCN forward at world sizes 2, 4 and 8 on ONE GPU.

`tests/_loopback.py` runs G virtual ranks of `ctgcn_amd.snapshot_parallel` in one process: the product code is unchanged, only its
`dist` module is swapped for device copies.  That puts the parts RCCL world 1 degenerates and gloo never sees under test on the GPU:
`send` written in place by the last CoreDiffusion, `recv [per, G, n_slice, d]`, the per-step offset table read by
`ops.gru_sequence_scattered`, node slices with padding (N not divisible by G), snapshot counts not divisible by G, ranks that own
nothing (G > T) — reference point `/root/reference/models.py:248-250`.  Every kernel treats rows as independent sequences, so the sharded
inference output must equal the unsharded HIP forward BIT FOR BIT; gradients to fp32 summation order."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _window(n, T, dev, model_type="C", hid=128):
    import ctgcn_amd
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import dynamic_graph
    graphs = dynamic_graph(n, 8, T, seed=9)
    adj = [core_adj_from_scipy(g, 4, dev)[0] for g in graphs]
    torch.manual_seed(0)
    if model_type == "C":
        model = ctgcn_amd.CTGCN(20, hid, 128, 1, 2, T).to(dev)
    else:
        model = ctgcn_amd.CTGCN(20, hid, 128, 2, 1, T, model_type="S", trans_activate_type="N").to(dev)
    torch.manual_seed(1)
    xs = [torch.randn(n, 20, device=dev) for _ in range(T)]
    return graphs, adj, model, xs


def _first(res):
    return res[0] if isinstance(res, tuple) else res


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
def test_sharded_inference_is_bit_identical_at_world(world, exchange, monkeypatch):
    from _loopback import LoopbackDist
    from ctgcn_amd import ops, snapshot_parallel as spp
    dev = torch.device("cuda", 0)
    n, T = 1001, 5                     # 1001 = 7 x 11 x 13: padded node slices at every world size; T = 5: uneven slots, idle ranks at G = 8
    graphs, adj, model, xs = _window(n, T, dev)
    model.eval()
    with torch.no_grad():
        want = model(xs, adj).clone()
    loop = LoopbackDist(world)
    monkeypatch.setattr(spp, "dist", loop)
    scattered = {"calls": 0}
    real = ops.gru_sequence_scattered

    def counting(*a, **k):
        scattered["calls"] += 1
        return real(*a, **k)
    monkeypatch.setattr(ops, "gru_sequence_scattered", counting)
    models = [copy.deepcopy(model) for _ in range(world)]
    costs = [g.nnz for g in graphs]

    def body(rank):
        m = models[rank]
        plan = spp.shard_ctgcn(m, n, costs=costs, group=loop.group.WORLD, exchange=exchange, gather_output=False)
        mine = plan.assignment[rank]
        lo, hi = plan.node_range(rank)
        x_l = [xs[t] if t in mine else None for t in range(T)]
        a_l = [adj[t] if t in mine else None for t in range(T)]
        rep = {"mine": mine, "range": (lo, hi), "per": plan.per, "n_slice": plan.n_slice}
        with torch.no_grad():
            got = m(x_l, a_l)
            again = m(x_l, a_l)                      # second forward: the cached exchange buffers (and their zeroed pads) are reused
        rep["shape_ok"] = tuple(got.shape) == (T, hi - lo, 128)
        rep["bitwise"] = bool(torch.equal(got, want[:, lo:hi]))
        rep["again"] = bool(torch.equal(again, got))
        rep["err"] = float((got - want[:, lo:hi]).abs().max()) if hi > lo else 0.0
        m.shard_gather_output = True
        with torch.no_grad():
            full = m(x_l, a_l)
        rep["gather_bitwise"] = bool(torch.equal(full, want))
        return rep
    reps = loop.run(body)
    owned = sorted(t for r in reps for t in r["mine"])
    assert owned == list(range(T))
    if world > T:
        assert any(not r["mine"] for r in reps)               # a rank that owns nothing still takes part
    assert sum(r["range"][1] - r["range"][0] for r in reps) == n and reps[-1]["range"][1] == n
    assert reps[0]["n_slice"] * world > n                     # padded
    for rank, r in enumerate(reps):
        assert r["shape_ok"] and r["bitwise"] and r["again"] and r["gather_bitwise"], (rank, r)
    if exchange == "all_to_all":
        # the pipelined path: one asynchronous all-to-all per slot, the temporal GRU reading recv through the offset table
        assert loop.calls["all_to_all_single"] == 3 * reps[0]["per"]
        assert scattered["calls"] == 3 * world


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("exchange", ["all_to_all", "all_gather"])
def test_sharded_autograd_branch_on_cuda_at_world(world, exchange, monkeypatch):
    """The training form (collectives as autograd Functions; backward = the transposed collective) on CUDA tensors: forward bit-identical,
    parameter gradients equal to the unsharded backward's up to fp32 summation order, replicated head gradients summed over ranks."""
    from _loopback import LoopbackDist
    from ctgcn_amd import snapshot_parallel as spp
    dev = torch.device("cuda", 0)
    n, T = 1001, 5
    graphs, adj, model, xs = _window(n, T, dev)
    gsel = torch.randn(T, n, 128, device=dev)
    model.train()
    out = model(xs, adj)
    want = out.detach().clone()
    (out * gsel).sum().backward()
    want_grad = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    loop = LoopbackDist(world)
    monkeypatch.setattr(spp, "dist", loop)
    models = [copy.deepcopy(model) for _ in range(world)]

    def body(rank):
        m = models[rank]
        plan = spp.shard_ctgcn(m, n, costs=[g.nnz for g in graphs], group=loop.group.WORLD, exchange=exchange, gather_output=False)
        mine = plan.assignment[rank]
        lo, hi = plan.node_range(rank)
        x_l = [xs[t] if t in mine else None for t in range(T)]
        a_l = [adj[t] if t in mine else None for t in range(T)]
        got = m(x_l, a_l)
        rep = {"fwd": bool(torch.equal(got.detach(), want[:, lo:hi]))}
        (got * gsel[:, lo:hi]).sum().backward()
        spp.allreduce_replicated_grads(m)
        owned = {id(p) for p in spp.owned_parameters(m)}
        worst, compared = 0.0, 0
        for k, p in m.named_parameters():
            if id(p) not in owned or k not in want_grad:
                continue
            assert p.grad is not None, k
            worst = max(worst, float((p.grad - want_grad[k]).abs().max()) / (1e-12 + float(want_grad[k].abs().max())))
            compared += 1
        rep["grad_rel_err"], rep["compared"] = worst, compared
        return rep
    reps = loop.run(body)
    for rank, r in enumerate(reps):
        assert r["fwd"], (rank, r)
        assert r["compared"] >= 6 and r["grad_rel_err"] <= 2e-5, (rank, r)
    assert loop.calls["all_reduce"] >= 6                        # the replicated temporal GRU / LayerNorm gradients


def test_sharded_inference_ctgcn_s_two_ranks(monkeypatch):
    """CTGCN-S (two transform layers, one diffusion layer, hid 500 in front of the 128-wide state): the structure outputs stay with their owner."""
    from _loopback import LoopbackDist
    from ctgcn_amd import snapshot_parallel as spp
    dev = torch.device("cuda", 0)
    n, T, world = 777, 4, 2
    graphs, adj, model, xs = _window(n, T, dev, model_type="S", hid=500)
    model.eval()
    with torch.no_grad():
        want, want_trans = model(xs, adj)
        want = want.clone()
    loop = LoopbackDist(world)
    monkeypatch.setattr(spp, "dist", loop)
    models = [copy.deepcopy(model) for _ in range(world)]

    def body(rank):
        m = models[rank]
        plan = spp.shard_ctgcn(m, n, group=loop.group.WORLD, gather_output=False)
        mine = plan.assignment[rank]
        lo, hi = plan.node_range(rank)
        with torch.no_grad():
            got, trans = m([xs[t] if t in mine else None for t in range(T)], [adj[t] if t in mine else None for t in range(T)])
        ok = bool(torch.equal(got, want[:, lo:hi]))
        for t in range(T):
            if t in mine:
                ok = ok and bool(torch.equal(trans[t], want_trans[t]))
            else:
                ok = ok and trans[t] is None
        return ok
    assert all(loop.run(body))


@pytest.mark.parametrize("rows,steps,ld_row", [(1000, 5, 128), (257, 16, 384), (33, 3, 128), (4096, 8, 256)])
def test_gru_sequence_scattered_reads_a_permuted_offset_table(rows, steps, ld_row):
    """ops.gru_sequence_scattered: step t of row r lives at base + step_offsets[t] + r * ld_row floats.  Steps laid out in a permuted order with
    gaps between them (and junk between rows when ld_row > 128) must give exactly the dense call's result."""
    from ctgcn_amd import ops
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    rnn = torch.nn.GRU(128, 128, batch_first=True).to(dev)
    norm = torch.nn.LayerNorm(128).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.2, 0.2)
    x = torch.randn(rows, steps, 128, device=dev)
    block = rows * ld_row + 64                                   # a step's rows, then 64 floats of junk before the next block
    base = torch.full((steps * block + 128,), float("nan"), device=dev)
    perm = torch.randperm(steps).tolist()
    offs = []
    for t in range(steps):
        off = 128 + perm[t] * block                              # 256-byte aligned starts (the receive buffer's slices are 512-byte aligned)
        offs.append(off)
        base[off: off + rows * ld_row].view(rows, ld_row)[:, :128] = x[:, t]
    step_off = torch.tensor(offs, dtype=torch.int64, device=dev)
    with torch.no_grad():
        assert ops.gru_steps_scattered_ok(rnn, base)
        want = ops.gru_sequence(rnn, x, norm, False)
        got = ops.gru_sequence_scattered(rnn, norm, base, step_off, ld_row, rows)
    assert got.shape == (rows, steps, 128)
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)
