#!/usr/bin/env python3
"""bench.py — CTGCN hot-path benchmark on MI355X (driver contract: one JSON line on rank 0).

step     = one embedding (forward) pass of CTGCN-C over the whole T-snapshot window: per-snapshot
           one-hot MLP -> 2 x CoreDiffusion (HIP aggregation + core-axis GRU + LayerNorm) -> exchange ->
           temporal GRU + LayerNorm.  Inputs (graphs, features, weights) are resident in HBM before timing.
metric   = aggregated edges/s over the window = sum_t sum_layers sum_k nnz(A(t,k)) / step time  (BASELINE.json)
workload = BASELINE config 5: synthetic power-law dynamic graph, 1M nodes x 16 cumulative snapshots,
           avg-deg 16, max_core capped at 8, hid = embed = 128.  Fits one GPU; with --gpus N the
           SAME window is sharded snapshot-parallel over N ranks (strong scaling).

Extra objects on the JSON line:
  roofline      dominant kernel (agg_fwd_kernel): algorithmic bytes per launch / HIP-event-measured duration
  cpu_baseline  the reference's torch.sparse.mm CPU loop (oracle/torch_path.py) on a bounded sample, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (nodes, snapshots, avg_deg, max_core, hid, embed, description)
    "synthetic-1m": (1_000_000, 16, 16, 8, 128, 128, "BASELINE config 5: synthetic 1M nodes x 16 snapshots, avg-deg 16"),
    "facebook-like": (60_730, 27, 20, 9, 128, 128, "BASELINE config 3 shape: 60 730 nodes x 27 snapshots (synthetic stand-in)"),
    "enron-like": (87_036, 12, 12, 5, 500, 128, "BASELINE config 2 shape: 87 036 nodes x 12 snapshots, max_core 5 (synthetic stand-in)"),
    "tiny": (20_000, 4, 8, 4, 64, 64, "debug"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="synthetic-1m", choices=sorted(WORKLOADS))
    ap.add_argument("--exchange", default="all_to_all", choices=["all_to_all", "all_gather"])
    ap.add_argument("--train", action="store_true",
                    help="time forward + backward + Adam step (surrogate loss out.square().mean(); the reference's "
                         "negative-sampling loss is outside the hot path) instead of the embedding forward")
    ap.add_argument("--graph", action="store_true",
                    help="replay the window from one captured hipGraph (ctgcn_amd.graph_capture; single GPU, inference): "
                         "removes launch/Python overhead on small graphs; per-launch HIP-event timing (roofline) is off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    return ap.parse_args()


def algorithmic_bytes(n, nnz, K, d):
    """SURVEY.md §8d, fused nested kernel: one pass over the largest matrix (4d B gathered row + 4 B col + 4 B val
    + 1 B slot per entry), K output rows of 4d B per node, row_ptr."""
    return nnz * (4 * d + 9) + n * K * 4 * d + 4 * (n + 1)


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: anything libraries print on fd 1 meanwhile (RCCL's version banner ...)
    # is routed to stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CTGCN_FORCE_DIST=1 runs the RCCL/sharded code path even with one rank (1-GPU boxes can exercise it)
    force_dist = os.environ.get("CTGCN_FORCE_DIST") == "1" and "RANK" in os.environ
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ctgcn_amd import CTGCN, CoreAdj, ops, _lib
    from ctgcn_amd import snapshot_parallel as spp
    from ctgcn_amd.synth import dynamic_graph_device, prefix_sizes, DEFAULT_SEED
    _lib.load()

    n, T, avg_deg, max_core, hid, emb, desc = WORKLOADS[args.workload]
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if rank == 0 else (lambda *a: None)

    # ---------------------------------------------------------------- plan + graphs (owned snapshots only)
    sizes = prefix_sizes(int(n * avg_deg / 2), T)
    assignment = spp.plan_assignment(sizes, world)
    mine = assignment[rank]
    t0 = time.time()
    graphs = dynamic_graph_device(n, avg_deg, T, dev, seed=DEFAULT_SEED, which=mine)
    log("generated %d/%d snapshots in %.1fs" % (len(mine), T, time.time() - t0))
    t0 = time.time()
    adj_list, local_stats = [None] * T, {}
    for t in mine:
        rp, col, val = graphs[t]
        adj, core, files = CoreAdj.from_graph(rp, col, val, max_core=max_core)
        adj_list[t] = adj
        local_stats[t] = dict(K=adj.K, nnz=adj.nnz, agg=adj.aggregated_edges, max_core=files)
    torch.cuda.synchronize()
    log("k-core + slot tagging of %d snapshots in %.2fs" % (len(mine), time.time() - t0))
    del graphs

    # every rank needs the window totals (metric numerator)
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, local_stats)
        stats = {}
        for g in gathered:
            stats.update(g)
    else:
        stats = local_stats
    layers = 2
    agg_edges_step = layers * sum(stats[t]["agg"] for t in range(T))

    # ------------------------------------------------------------------------------- model + features
    torch.manual_seed(0)
    with torch.device(dev):
        model = CTGCN(n, hid, emb, 1, layers, T, rnn_type="GRU", model_type="C", trans_activate_type="L")
    model.eval()
    eye_idx = torch.arange(n, device=dev).repeat(2, 1)
    x_list = [None] * T
    for t in mine:   # one-hot node features = sparse identity (reference helper.py:161-172)
        x_list[t] = torch.sparse_coo_tensor(eye_idx, torch.ones(n, device=dev), (n, n))
    if use_dist:
        spp.shard_ctgcn(model, n, assignment=assignment, exchange=args.exchange, gather_output=False)

    # HIP-event timing of every aggregation launch (same stream the kernel is launched on)
    launches = []
    if args.graph:
        if use_dist or args.train:
            raise SystemExit("--graph: single-GPU inference only")
        from ctgcn_amd.graph_capture import GraphedInference
        runner = GraphedInference(model, x_list, adj_list)
    else:
        ops.set_launch_timer(lambda name, start, end, meta: launches.append((name, start, end, meta)))

    if args.train:
        model.train()
        params = list(spp.owned_parameters(model)) if use_dist else list(model.parameters())
        opt = torch.optim.Adam(params, lr=1e-3)

    def step():
        if args.graph:
            return runner()
        if not args.train:
            with torch.no_grad():
                return model(x_list, adj_list)
        opt.zero_grad(set_to_none=True)
        out = model(x_list, adj_list)
        out.square().mean().backward()
        if use_dist:
            spp.allreduce_replicated_grads(model)
        opt.step()
        return out.detach()

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    launches.clear()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t_start
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1000.0 * elapsed / args.steps
    assert torch.isfinite(out).all()

    # ------------------------------------------------------------------------- roofline of the dominant kernel
    ops.set_launch_timer(None)
    fwd = [(s.elapsed_time(e), meta) for name, s, e, meta in launches if name == "agg_fwd"]
    kern_ms = [ms for ms, _ in fwd]
    kern_bytes = [algorithmic_bytes(m["n"], m["nnz"], m["K"], m["d"]) for _, m in fwd]
    roof = None
    if fwd:
        avg_ms = sum(kern_ms) / len(kern_ms)
        avg_bytes = sum(kern_bytes) / len(kern_bytes)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_file):
            try:
                rec = json.load(open(pmc_file)).get(args.workload, {}).get(str(world))
                traffic = rec["hbm_bytes_per_launch"] if rec else None
            except Exception:
                traffic = None
        roof = {"kernel": "agg_fwd_kernel<4,32,4> (CoreDiffusion fused nested-core SpMM, d=128)", "bound": "hbm",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "launches_timed": len(fwd), "avg_launch_ms": round(avg_ms, 4),
                "algorithmic_bytes_per_launch": int(avg_bytes),
                "frac_of_measured_copy_bw_6300": round(achieved / 6300.0, 4)}
    spmm_ms_step = sum(kern_ms) / args.steps if kern_ms else None
    # the matrix-core kernels: GRU recurrence (+ sum/LayerNorm) and the input projection
    gru = [(s.elapsed_time(e), meta) for name, s, e, meta in launches if name == "gru_seq"]
    proj = [(s.elapsed_time(e), meta) for name, s, e, meta in launches if name == "gru_proj"]
    roof_mfma = None
    if gru:
        mode = ops.forward_split_mode()
        # fp32-equivalent flops; step 0 (h = 0) issues no MFMA.  The matrix-core bound is the dense 16-bit peak divided by
        # the number of 16-bit products per fp32 product (fp16x2: 3, bf16x3: 6), or the fp32 MFMA peak for the exact path.
        peak = {2: 2500.0 / 3.0, 1: 2500.0 / 6.0, 0: 157.3}[mode]
        name = {2: "gru_seq_h2_kernel (GRU recurrence + sum + LayerNorm; fp32 operands scaled per row and split into two fp16 "
                   "terms, 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate)",
                1: "gru_seq_x3_kernel (GRU recurrence + sum + LayerNorm; fp32 operands split 3-way into bf16, "
                   "6 x v_mfma_f32_16x16x32_bf16 per product, fp32 accumulate)",
                0: "gru_seq_kernel (GRU recurrence + sum + LayerNorm, v_mfma_f32_16x16x4_f32)"}[mode]
        pname = {2: "gru_proj_h2_kernel", 1: "gru_proj_x3_kernel", 0: "hipBLASLt fp32 GEMM"}[mode]
        flops = sum(m["rows"] * (m["steps"] - 1) * 2.0 * 128 * 384 for _, m in gru)
        gi_bytes = sum(m["rows"] * m["steps"] * 1536.0 for _, m in gru)    # the projection's output read back, 1536 B per row-step
        ms = sum(t for t, _ in gru)
        roof_mfma = {"kernel": name,
                     "bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 2), "peak": round(peak, 1),
                     "unit": "TFLOP/s (fp32-equivalent)", "frac": round(flops / (ms * 1e-3) / 1e12 / peak, 4),
                     "gi_read_GBps": round(gi_bytes / (ms * 1e-3) / 1e9, 1),
                     "launches_timed": len(gru), "ms_per_step_rank0": round(ms / args.steps, 3)}
        if proj:
            pms = sum(t for t, _ in proj)
            pbytes = sum(m["rows"] * 2048.0 for _, m in proj)          # 512 B read + 1536 B written per row
            roof_mfma["input_projection"] = {"kernel": pname, "bound": "hbm", "ms_per_step_rank0": round(pms / args.steps, 3),
                                             "achieved": round(pbytes / (pms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": round(pbytes / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if use_dist:
        dist.barrier()
    if rank != 0:
        dist.destroy_process_group()
        return

    # ------------------------------------------------------------------------------------- CPU baseline (N=1)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(adj_list[T - 1], emb, args.cpu_budget_s, log)

    line = {
        "metric": "aggregated edges/s over T-snapshot window (CTGCN-C %s)" % ("training step: forward + backward + Adam" if args.train else "embedding forward"),
        "value": agg_edges_step / (ms_per_step * 1e-3),
        "unit": "edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seed %d power-law dynamic graph, random-init weights)" % DEFAULT_SEED,
        "config": {"workload": "%s; CTGCN-C hid=%d embed=%d, 2 diffusion layers, max_core=%d" % (desc, hid, emb, max_core),
                   "nodes": n, "snapshots": T, "avg_deg": avg_deg, "max_core": max_core,
                   "K_per_snapshot": [stats[t]["K"] for t in range(T)],
                   "stored_entries_per_snapshot": [stats[t]["nnz"] for t in range(T)],
                   "aggregated_edges_per_step": agg_edges_step,
                   "parallelism": "snapshot-parallel x%d (%s exchange before the temporal GRU)" % (world, args.exchange) if world > 1 else ("single GPU, hipGraph replay" if args.graph else "single GPU"),
                   "assignment": assignment},
        "embed_wall_ms": round(ms_per_step, 3),
        "aggregation_ms_per_step_rank0": None if spmm_ms_step is None else round(spmm_ms_step, 3),
        "aggregation_edges_per_s_rank0": None if not spmm_ms_step else
            layers * sum(stats[t]["agg"] for t in mine) / (spmm_ms_step * 1e-3),
        "roofline": roof,
        "roofline_gru": roof_mfma,
        "cpu_baseline": cpu,
    }
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


def cpu_baseline(adj, d, budget_s, log):
    """Reference CPU path (layers.py:41-48 loop of torch.sparse.mm on uncoalesced COO, utils.py:89-95) timed on
    the host cores for ONE CoreDiffusion aggregation of the window's LAST snapshot at d=128; stops adding
    matrices once the budget is spent and reports edges/s over what was run."""
    from oracle import torch_path as TP
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    mats = adj.cpu().to_scipy_list()
    coo = [TP.coo_like_reference(m) for m in mats]
    x = torch.randn(adj.n, d)
    done_edges, t_used, acc, used = 0, 0.0, None, 0
    for j, a in enumerate(coo):
        t0 = time.perf_counter()
        y = torch.sparse.mm(a, x)
        acc = y if acc is None else acc + y
        acc_r = torch.relu(acc)
        dt = time.perf_counter() - t0
        t_used += dt
        done_edges += a._nnz()
        used = j + 1
        if t_used > budget_s:
            break
    del acc_r
    log("cpu baseline: %d/%d matrices, %.1fs" % (used, len(coo), t_used))
    return {"value": done_edges / t_used, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": "oracle/torch_path.py (reference layers.py:41-48 restated: torch.sparse.mm on uncoalesced COO + add + relu), "
                      "last snapshot of the window, first %d of %d k-core matrices, d=%d, %d aggregated edges in %.1f s, "
                      "single un-warmed pass" % (used, len(coo), d, done_edges, t_used)}


if __name__ == "__main__":
    main()
