#!/usr/bin/env python3
"""Full-array parity of the HIP path at the BASELINE config sizes against the float64 CPU oracle (VERDICT r5 "What's weak" 1 / "do this" 3).

The GPU suite (tests/test_gpu_configs.py) evaluates the float64 oracle on a node sample to stay inside its time limit; a tail fraction of
~1e-5 estimated on 1e6 sampled entries is a count of a few dozen, so the suite's numbers cannot say whether the HIP path's fraction of
entries outside rtol 1e-4 / atol 1e-5 is 1.1x or 1.5x the fp32 CPU path's.  This tool evaluates the oracle in float64 on EVERY row, once
per round, outside the suite:

    gpurun --timeout 1500 -- 'python tools/parity_full.py --out gpurun_out/parity_full.json'        (then copy to profiles/rNN_parity_full.json)

Per case: entries, entries outside the tolerance for the HIP path and for the fp32 CPU path (the reference's arithmetic), their ratio
with a 95 % interval (Poisson bootstrap over NODES — the two counts are paired and errors cluster in hub rows, so entries are not
independent draws), the worst and RMS errors of both against float64.  The oracle here is the checker (test infrastructure); nothing of
it is in the product."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# the outlier-fraction slack per case (tests/test_gpu_configs.py: FRAC_SLACK; the 16-step window's own, set from round 6's interval 1.18 - 1.50)
SLACK = {"window_T16_n200k": 1.6}


def stats(got, want, want64, seed=0, boots=2000, degree=None):
    """got / want (fp32 HIP, fp32 CPU oracle) / want64: [T, n, d] arrays; degree [n]: stored entries per node of the largest snapshot (where the
    outliers live: by step, and by the degree rank of their node)"""
    tol = 1e-4 * np.abs(want64) + 1e-5
    d_hip, d_cpu = np.abs(got - want64), np.abs(want - want64)
    bad_h, bad_c = d_hip > tol, d_cpu > tol
    per_node_h = bad_h.sum(axis=(0, 2)).astype(np.float64)
    per_node_c = bad_c.sum(axis=(0, 2)).astype(np.float64)
    nz = np.flatnonzero((per_node_h + per_node_c) > 0)
    h, c = per_node_h[nz], per_node_c[nz]
    rng = np.random.default_rng(seed)
    ratios = []
    if c.sum() > 0:
        for _ in range(boots):
            w = rng.poisson(1.0, len(nz))
            den = float((w * c).sum())
            if den > 0:
                ratios.append(float((w * h).sum()) / den)
    lo, hi = (float(np.percentile(ratios, 2.5)), float(np.percentile(ratios, 97.5))) if ratios else (None, None)
    size = got.size
    where = {"outside_by_step_hip": [int(x) for x in bad_h.sum(axis=(1, 2))], "outside_by_step_cpu_fp32": [int(x) for x in bad_c.sum(axis=(1, 2))]}
    if degree is not None and len(nz):
        order = np.argsort(-np.asarray(degree), kind="stable")
        rank = np.empty(len(order), dtype=np.int64)
        rank[order] = np.arange(len(order))
        top1 = rank[nz] < max(1, len(order) // 100)
        where.update(outlier_nodes_in_top_1pct_degree=int(top1.sum()), outside_hip_in_those=int(per_node_h[nz][top1].sum()),
                     outside_cpu_in_those=int(per_node_c[nz][top1].sum()), median_degree_of_outlier_nodes=float(np.median(np.asarray(degree)[nz])),
                     median_degree_overall=float(np.median(degree)))
    return {"entries": int(size), "nodes_with_an_outlier": int(len(nz)), "where": where,
            "outside_hip": int(bad_h.sum()), "outside_cpu_fp32": int(bad_c.sum()), "outside_both": int((bad_h & bad_c).sum()),
            "frac_outside_hip": float(bad_h.sum()) / size, "frac_outside_cpu_fp32": float(bad_c.sum()) / size,
            "ratio_hip_over_cpu": (float(bad_h.sum()) / float(bad_c.sum())) if bad_c.sum() else None, "ratio_95_interval": [lo, hi],
            "max_err_hip_vs_fp64": float(d_hip.max()), "max_err_cpu_fp32_vs_fp64": float(d_cpu.max()),
            "rms_err_hip_vs_fp64": float(np.sqrt(np.mean(d_hip.astype(np.float64) ** 2))),
            "rms_err_cpu_fp32_vs_fp64": float(np.sqrt(np.mean(d_cpu.astype(np.float64) ** 2))),
            "mean_err_hip_vs_fp64": float(d_hip.mean()), "mean_err_cpu_fp32_vs_fp64": float(d_cpu.mean()),
            "max_err_hip_vs_fp32_oracle": float(np.abs(got - want).max()),
            "rule": "|x - x64| > 1e-4 |x64| + 1e-5 counts as outside (SURVEY 8c's tolerance after GRU + LayerNorm)"}


def small_case(case):
    import ctgcn_amd
    import test_gpu_configs as G
    c, graphs, adj, ref_adj, mats = G._build(case)
    xs, input_dim = G._features(c, graphs)
    torch.manual_seed(0)
    model = ctgcn_amd.CTGCN(input_dim, c["hid"], 128, c["trans"], c["diff"], c["T"], rnn_type="GRU", model_type=c["model"], trans_activate_type=c["act"]).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(G.DEV)
    with torch.no_grad():
        got = model([x.to(G.DEV) for x in xs], adj)
    want, want64, t32, t64 = G._oracle_fp32_and_fp64(sd, xs, ref_adj, "GRU", c["model"], c["act"])
    if c["model"] == "S":
        got, want, want64 = got[0], want[0], want64[0]
    out = stats(got.cpu().numpy(), want.numpy(), want64.numpy())
    out.update(oracle_fp32_s=round(t32, 1), oracle_fp64_s=round(t64, 1), nodes=c["n"], snapshots=c["T"])
    return out


def config5():
    """snapshots 3 and 15 of the 1 M-node window, as tests/test_gpu_configs.py::config5 builds them; inference path and autograd forward"""
    import ctgcn_amd
    import scipy.sparse as sp
    import test_gpu_configs as G
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import snapshot_rows
    from oracle import oracle as O, torch_path as TP
    C5 = G.C5
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
        a, core_dev, _ = core_adj_from_scipy(g, K, G.DEV)
        capped = np.minimum(core, K)
        assert np.array_equal(core_dev.cpu().numpy(), capped)
        ref = O.core_adj_list([O.kcore_matrices(g, capped)], 0, 1, 1, max_core=K)[0]
        adj.append(a)
        ref_adj.append([TP.coo_like_reference(m) for m in ref])
    idx = torch.arange(n).repeat(2, 1)
    xs = [torch.sparse_coo_tensor(idx, torch.ones(n), (n, n)) for _ in graphs]
    torch.manual_seed(0)
    model = ctgcn_amd.CTGCN(n, 128, 128, 1, 2, len(graphs)).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(G.DEV)
    xd = [x.to(G.DEV) for x in xs]
    with torch.no_grad():
        got_inf = model(xd, adj).cpu().numpy()
    model.train()
    got_train = model(xd, adj).detach().cpu().numpy()
    model.eval()
    want, want64, t32, t64 = G._oracle_fp32_and_fp64(sd, xs, ref_adj)
    want, want64 = want.numpy(), want64.numpy()
    res = {}
    for name, got in (("config5_full_inference", got_inf), ("config5_full_autograd_forward", got_train)):
        res[name] = stats(got, want, want64)
        res[name].update(oracle_fp32_s=round(t32, 1), oracle_fp64_s=round(t64, 1), nodes=n, snapshots=list(C5["pick"]), K=[len(a) for a in adj])
    # the exact-fp32 build of the same forward (CTGCN_FP32_MFMA_ONLY=1: no 16-bit operand anywhere) — where the excess over the CPU path comes from
    os.environ["CTGCN_FP32_MFMA_ONLY"] = "1"
    try:
        with torch.no_grad():
            got_exact = model(xd, adj).cpu().numpy()
    finally:
        del os.environ["CTGCN_FP32_MFMA_ONLY"]
    res["config5_full_inference_exact_fp32_mode"] = stats(got_exact, want, want64)
    return res


def window_t16():
    """the T = 16 window at 200 000 nodes of tests/test_gpu_configs.py::test_full_depth_window_matches_cpu_oracle_on_sampled_rows, every row"""
    import ctgcn_amd
    import test_gpu_configs as G
    from ctgcn_amd.helper import core_adj_from_scipy
    from ctgcn_amd.synth import window_graph
    from oracle import oracle as O, torch_path as TP
    n, T, K = 200_000, 16, 8
    graphs = window_graph(n, 1_600_000, T, cumulative=True)
    adj, ref_adj = [], []
    for g in graphs:
        capped = np.minimum(O.core_numbers(g), K)
        a, core_dev, _ = core_adj_from_scipy(g, K, G.DEV)
        assert np.array_equal(core_dev.cpu().numpy(), capped)
        ref = O.core_adj_list([O.kcore_matrices(g, capped)], 0, 1, 1, max_core=K)[0]
        adj.append(a)
        ref_adj.append([TP.coo_like_reference(m) for m in ref])
    idx = torch.arange(n).repeat(2, 1)
    xs = [torch.sparse_coo_tensor(idx, torch.ones(n), (n, n)) for _ in range(T)]
    torch.manual_seed(0)
    model = ctgcn_amd.CTGCN(n, 128, 128, 1, 2, T).eval()
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(G.DEV)
    with torch.no_grad():
        got = model([x.to(G.DEV) for x in xs], adj).cpu().numpy()
    os.environ["CTGCN_FP32_MFMA_ONLY"] = "1"           # the same forward without any 16-bit operand (hub rows have a wide dynamic range per row)
    try:
        with torch.no_grad():
            got_exact = model([x.to(G.DEV) for x in xs], adj).cpu().numpy()
    finally:
        del os.environ["CTGCN_FP32_MFMA_ONLY"]
    want, want64, t32, t64 = G._oracle_fp32_and_fp64(sd, xs, ref_adj)
    deg = np.diff(graphs[-1].indptr)
    out = stats(got, want.numpy(), want64.numpy(), degree=deg)
    out.update(oracle_fp32_s=round(t32, 1), oracle_fp64_s=round(t64, 1), nodes=n, snapshots=T)
    ex = stats(got_exact, want.numpy(), want64.numpy(), degree=deg)
    out["exact_fp32_mode"] = {k: ex[k] for k in ("outside_hip", "outside_cpu_fp32", "ratio_hip_over_cpu", "ratio_95_interval", "rms_err_hip_vs_fp64",
                                                  "max_err_hip_vs_fp64", "where")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_full.json"))
    ap.add_argument("--cases", default="config5,math_c4,enron_c2,as_c4,facebook_s_c3,window_T16_n200k")
    args = ap.parse_args()
    assert torch.cuda.is_available()
    res = {"what": "full-array float64 parity (tools/parity_full.py); every row of every snapshot", "torch": torch.__version__}
    for case in args.cases.split(","):
        t0 = time.time()
        if case == "config5":
            res.update(config5())
        elif case == "window_T16_n200k":
            res[case] = window_t16()
        else:
            res[case] = small_case(case)
        print("%s done in %.0f s" % (case, time.time() - t0), flush=True)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1, sort_keys=True)       # after every case: a time-out keeps what was finished
    failed = []
    for k, v in res.items():
        if isinstance(v, dict):
            # the plain rule of tests/test_gpu_configs.py on full arrays: no sampling allowance
            slack = SLACK.get(k, 1.25)
            v["rule_applied"] = "outside_hip <= %.2f x outside_cpu_fp32 + 1e-6 x entries; worst <= 1.5 x + 2e-6; <= 5e-4 from the fp32 oracle" % slack
            if v["outside_hip"] > slack * v["outside_cpu_fp32"] + 1e-6 * v["entries"] or v["max_err_hip_vs_fp64"] > 1.5 * v["max_err_cpu_fp32_vs_fp64"] + 2e-6 \
                    or v["max_err_hip_vs_fp32_oracle"] > 5e-4:
                failed.append(k)
            print("%-44s outside HIP %8d / CPU %8d of %.2e  ratio %s  95%% %s  rms %.2e / %.2e  worst %.2e / %.2e" % (
                k, v["outside_hip"], v["outside_cpu_fp32"], v["entries"], v["ratio_hip_over_cpu"] and round(v["ratio_hip_over_cpu"], 3),
                [x and round(x, 3) for x in v["ratio_95_interval"]], v["rms_err_hip_vs_fp64"], v["rms_err_cpu_fp32_vs_fp64"],
                v["max_err_hip_vs_fp64"], v["max_err_cpu_fp32_vs_fp64"]))

    json.dump(res, open(args.out, "w"), indent=1, sort_keys=True)
    if failed:
        raise SystemExit("full-array parity rule broken: %s" % ", ".join(failed))


if __name__ == "__main__":
    main()
