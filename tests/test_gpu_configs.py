"""BASELINE configs 2, 3, 4 on the GPU (VERDICT r1 "configs untested"): the HIP path on the Enron / Facebook-S / math / AS
window shapes (synthetic stand-ins with the datasets' published statistics, reference README.md:168-176) against the CPU
oracle (oracle/torch_path.py: the reference's torch.sparse path, pinned to reference outputs by tests/test_oracle_golden.py)
on the FULL output, plus the loader-level integers (K per snapshot, core numbers) bit for bit.

Tolerance.  SURVEY.md §8c gives rtol 1e-4 / atol 1e-5 after GRU + LayerNorm, probed on UCI (1 899 nodes, degree <= 198).  At
these sizes (10^8 output values, hub rows that sum hundreds of 500-wide terms, degree features of magnitude 10^2-10^3 through
three SELU layers, three stacked recurrences) the fp32 CPU path ITSELF is further than that from the exact result on a small
fraction of the entries (facebook shape: 3.7e-4 worst).  The oracle is therefore also evaluated in float64 and the HIP path is
held to the fp32 reference path's OWN distance from that exact result:
  (a) the fraction of entries outside rtol 1e-4 / atol 1e-5 of the float64 result is at most FRAC_SLACK x the fp32 CPU path's
      fraction (+ 1e-6), and against the fp32 oracle itself the HIP output is nowhere further than 5e-4;
  (b) the worst HIP error against float64 is at most WORST_SLACK x the fp32 CPU path's worst error (+ 2e-6).
Observed errors are printed per case and written to gpurun_out/parity_errors/<case>.json (committed copy of a full run:
profiles/r03_parity_errors.json).

Config 5 (1 M nodes) is held to the same rule at FULL size (test_config5_full_size_matches_cpu_oracle): snapshots 3 and 15 of the
16-snapshot window, max_core 8, through the inference path (aggregation -> fp16 planes -> register-resident GRU layer kernel), the
autograd forward (fp32 rows) and the hub-row kernels (LONG_ROW forced low)."""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FRAC_SLACK = 1.25       # observed <= 1.16 (profiles/r03_parity_errors.json)
WORST_SLACK = 1.5       # observed <= 1.40: the single worst of up to 2e8 entries is an extreme-value statistic; round 2 allowed 2.0


def _record(case, **numbers):
    """observed errors -> gpurun_out/parity_errors/<case>.json (scratch; a full run is committed as profiles/r03_parity_errors.json)"""
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_errors")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, case + ".json"), "w") as f:
            json.dump({k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in numbers.items()}, f, indent=1, sort_keys=True)
    except OSError:
        pass


def _oracle_fp32_and_fp64(sd, xs, ref_adj, *model_args):
    """the CPU oracle in fp32 (the reference's path: uncoalesced COO operands) and in fp64 (the truth both sides are measured against;
    operands coalesced once — their order is immaterial there).  One after the other: run concurrently the two oversubscribe the host's
    cores and take 1.4x longer in total (measured)."""
    from oracle import torch_path as TP
    t0 = time.time()
    with torch.no_grad():
        want = TP.ctgcn(sd, xs, ref_adj, *model_args)
        t32 = time.time() - t0
        want64 = TP.ctgcn({k: v.double() for k, v in sd.items()}, [x.double() for x in xs], [[a.double().coalesce() for a in l] for l in ref_adj], *model_args)
    return want, want64, t32, time.time() - t0 - t32


def _compare(case, got, want, want64, extra=None):
    """the rule of the module docstring on full output arrays; returns the observed numbers"""
    err = np.abs(got - want)
    tol64 = 1e-4 * np.abs(want64) + 1e-5
    d_hip, d_cpu = np.abs(got - want64), np.abs(want - want64)
    bad_hip, bad_cpu = float((d_hip > tol64).mean()), float((d_cpu > tol64).mean())
    err_hip64, err_cpu64 = float(d_hip.max()), float(d_cpu.max())
    obs = dict(max_err_vs_fp32_oracle=float(err.max()), mean_err_vs_fp32_oracle=float(err.mean()), max_err_hip_vs_fp64=err_hip64,
               max_err_cpu_fp32_vs_fp64=err_cpu64, frac_outside_hip=bad_hip, frac_outside_cpu_fp32=bad_cpu,
               rule="rtol 1e-4 atol 1e-5 vs fp64; slack %.2f / %.2f" % (FRAC_SLACK, WORST_SLACK), shape=list(got.shape))
    obs.update(extra or {})
    print("%s: vs fp32 oracle max |err| %.3e mean %.3e; vs fp64 oracle: max HIP %.3e / fp32 CPU path %.3e, fraction outside rtol 1e-4 atol 1e-5 "
          "HIP %.2e / fp32 CPU path %.2e" % (case, err.max(), err.mean(), err_hip64, err_cpu64, bad_hip, bad_cpu))
    _record(case, **obs)
    assert bad_hip <= FRAC_SLACK * bad_cpu + 1e-6 and err.max() <= 5e-4, obs
    assert err_hip64 <= WORST_SLACK * err_cpu64 + 2e-6, obs
    return obs

CASES = {
    # name: nodes, edges, T, cumulative, max_core, hid, model, trans, diff, act, features
    "enron_c2": dict(n=87_036, edges=530_284, T=12, cumulative=True, max_core=5, hid=500, model="C", trans=1, diff=2, act="L", feat="one-hot"),
    "facebook_s_c3": dict(n=60_730, edges=607_487, T=27, cumulative=True, max_core=-1, hid=500, model="S", trans=3, diff=1, act="N", feat="gaussian"),
    "math_c4": dict(n=24_740, edges=323_357, T=8, cumulative=True, max_core=-1, hid=500, model="C", trans=1, diff=2, act="L", feat="one-hot"),
    "as_c4": dict(n=6_828, edges=19_500, T=8, cumulative=False, max_core=-1, hid=500, model="C", trans=1, diff=2, act="L", feat="one-hot"),
}


def _build(case):
    """graphs on the host (oracle side) and through the device route (product side), with the loader's sticky max_core"""
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import window_graph
    from oracle import oracle as O, torch_path as TP
    c = CASES[case]
    graphs = window_graph(c["n"], c["edges"], c["T"], cumulative=c["cumulative"], max_degree_hint=500 if c["feat"] == "gaussian" else None)
    cores = [O.core_numbers(g) for g in graphs]
    mats = [O.kcore_matrices(g, core) for g, core in zip(graphs, cores)]
    ref_lists = O.core_adj_list(mats, 0, c["T"], c["T"], max_core=c["max_core"])
    adj, mc = [], c["max_core"]
    for g, core in zip(graphs, cores):
        a, core_dev, files = core_adj_from_scipy(g, mc, DEV)
        want = core if mc < 0 else np.minimum(core, mc)
        assert np.array_equal(core_dev.cpu().numpy(), want)                  # integer k-core assignment: bit-exact
        if mc == -1:
            mc = files                                                       # helper.py:61-62
        adj.append(a)
    assert [len(a) for a in adj] == [len(l) for l in ref_lists]
    for a, l in zip(adj, ref_lists):
        assert a.nnz_per_slot == [m.nnz for m in l]
    ref_adj = [[TP.coo_like_reference(m) for m in l] for l in ref_lists]
    return c, graphs, adj, ref_adj


def _features(c, graphs):
    n, T = c["n"], c["T"]
    if c["feat"] == "one-hot":
        idx = torch.arange(n).repeat(2, 1)
        xs = [torch.sparse_coo_tensor(idx, torch.ones(n), (n, n)) for _ in range(T)]
        return xs, n
    # Normal(degree, 1e-4) rows of width max degree + 1 (helper.py:128-135); the SAME host tensors feed both sides
    deg = [np.asarray(g.sum(axis=1)).reshape(-1).astype(int) for g in graphs]
    width = max(int(d.max()) for d in deg) + 1
    rng = np.random.default_rng(7)
    xs = [torch.from_numpy((d[:, None] + 1e-4 * rng.standard_normal((n, width))).astype(np.float32)) for d in deg]
    return xs, width


@pytest.mark.parametrize("case", sorted(CASES))
def test_baseline_config_shapes_match_cpu_oracle(case):
    import ctgcn_amd
    from ctgcn_amd import ops
    from oracle import torch_path as TP
    c, graphs, adj, ref_adj = _build(case)
    xs, input_dim = _features(c, graphs)
    torch.manual_seed(0)
    model = ctgcn_amd.CTGCN(input_dim, c["hid"], 128, c["trans"], c["diff"], c["T"], rnn_type="GRU", model_type=c["model"],
                            trans_activate_type=c["act"]).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV)
    assert ops.gru_fused_ok(model.rnn, torch.zeros(1, 1, 128, device=DEV))     # the HIP GRU kernels are the ones running
    with torch.no_grad():
        got = model([x.to(DEV) for x in xs], adj)
    want, want64, t32, t64 = _oracle_fp32_and_fp64(sd, xs, ref_adj, "GRU", c["model"], c["act"])
    if c["model"] == "S":
        (got, got_tr), (want, want_tr), (want64, _) = got, want, want64
        for a, b in zip(got_tr, want_tr):
            scale = float(b.abs().max())
            assert float((a.cpu() - b).abs().max()) <= 1e-5 * scale + 1e-6, "transform outputs (dense Linear + SELU)"
    got = got.cpu().numpy()
    want, want64 = want.numpy(), want64.numpy()
    assert got.shape == want.shape == (c["T"], c["n"], 128)
    _compare(case, got, want, want64, dict(oracle_fp32_s=t32, oracle_fp64_s=t64))


# ------------------------------------------------------------------------------------------ config 5 at full size
C5 = dict(n=1_000_000, edges=8_000_000, T=16, max_core=8, pick=(3, 15))


@pytest.fixture(scope="module")
def config5():
    """Snapshots 3 and 15 of bench.py's synthetic-1m window (same generator, same seed), the device-route CoreAdj of each, the
    reference-shaped COO lists, a seeded CTGCN-C(1 M one-hot, 128, 128, 1, 2, T = 2) and the CPU oracle's fp32 / fp64 outputs."""
    import ctgcn_amd
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import snapshot_rows
    from oracle import oracle as O, torch_path as TP
    import scipy.sparse as sp
    n, K = C5["n"], C5["max_core"]
    u, v, picks = snapshot_rows(n, C5["edges"], C5["T"], cumulative=True)
    graphs = []
    for t in C5["pick"]:
        uu, vv = u[picks[t]], v[picks[t]]
        a = sp.coo_matrix((np.ones(2 * len(uu)), (np.concatenate([uu, vv]), np.concatenate([vv, uu]))), shape=(n, n)).tocsr()
        a.sort_indices()
        graphs.append(a)
    adj, ref_adj = [], []
    for g in graphs:
        core = O.core_numbers(g)
        a, core_dev, _ = core_adj_from_scipy(g, K, DEV)
        capped = np.minimum(core, K)
        assert np.array_equal(core_dev.cpu().numpy(), capped)                   # integer k-core assignment: bit-exact at 1 M nodes
        ref = O.core_adj_list([O.kcore_matrices(g, capped)], 0, 1, 1, max_core=K)[0]      # levels above max_core are never told apart (helper.py:63)
        assert a.nnz_per_slot == [m.nnz for m in ref] and len(a) == len(ref)
        adj.append(a)
        ref_adj.append([TP.coo_like_reference(m) for m in ref])
    idx = torch.arange(n).repeat(2, 1)
    xs = [torch.sparse_coo_tensor(idx, torch.ones(n), (n, n)) for _ in graphs]
    torch.manual_seed(0)
    model = ctgcn_amd.CTGCN(n, 128, 128, 1, 2, len(graphs)).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(DEV)
    want, want64, t32, t64 = _oracle_fp32_and_fp64(sd, xs, ref_adj)
    return dict(model=model, adj=adj, xs=[x.to(DEV) for x in xs], want=want.numpy(), want64=want64.numpy(),
                times=dict(oracle_fp32_s=t32, oracle_fp64_s=t64), K=[len(a) for a in adj], nnz=[a.nnz for a in adj])


@pytest.mark.parametrize("path", ["inference", "autograd_forward", "hub_rows"])
def test_config5_full_size_matches_cpu_oracle(config5, path):
    """the fused inference path had never met the oracle above 87 036 rows (and a layer-kernel bug once was
    invisible at 70 001 rows, non-finite at 1 M).  Reference chain: models.py:240-253 -> layers.py:38-63."""
    from ctgcn_amd.core_adj import CoreAdj
    c = config5
    model, old = c["model"], CoreAdj.LONG_ROW
    extra = dict(c["times"], K=c["K"], nnz=c["nnz"], path=path)
    try:
        if path == "hub_rows":
            CoreAdj.LONG_ROW = 96          # ~1 % of the rows of snapshot 15 go through the block-per-row kernels (+ the mapped split)
            for a in c["adj"]:
                a._long.clear()
            extra["hub_rows"] = [0 if a.long_rows() is None else int(a.long_rows().numel()) for a in c["adj"]]
            assert min(extra["hub_rows"]) > 1000
        if path == "autograd_forward":
            model.train()
            got = model(c["xs"], c["adj"])
            assert got.requires_grad
            got = got.detach()
        else:
            model.eval()
            with torch.no_grad():
                got = model(c["xs"], c["adj"])
    finally:
        CoreAdj.LONG_ROW = old
        for a in c["adj"]:
            a._long.clear()
        model.eval()
    got = got.cpu().numpy()
    assert got.shape == c["want"].shape == (2, C5["n"], 128) and np.isfinite(got).all()
    _compare("config5_full_" + path, got, c["want"], c["want64"], extra)
