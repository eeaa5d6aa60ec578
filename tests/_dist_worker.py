"""Worker for tests/test_snapshot_parallel_gloo.py (spawned, world_size 2, gloo, CPU).

The HIP aggregation cannot run here, so the worker injects the CPU oracle at the product's single
dispatch point (ctgcn_amd.ops.core_aggregate) — test infrastructure acting as the checker; what is under
test is the sharding plan, the exchange collectives and their autograd, not the kernel."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _oracle_aggregate(x, adj, relu=True):
    from oracle import torch_path as TP
    mats = [TP.coo_like_reference(m) for m in adj.to_scipy_list()]
    hs = TP.aggregate_loop(mats, x) if relu else None
    return torch.stack(hs, 0).transpose(0, 1)


def run(rank, world, port, T, n, exchange, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctgcn_amd
        from ctgcn_amd import CoreAdj, ops, snapshot_parallel as spp
        from ctgcn_amd.synth import dynamic_graph
        from oracle import oracle as O, torch_path as TP
        from conftest import formula_tensor
        ops.core_aggregate = _oracle_aggregate

        graphs = dynamic_graph(n, 6, T, seed=11)
        lists = [O.core_adj_list([O.kcore_matrices(g)], 0, 1, 1, max_core=3)[0] for g in graphs]
        costs = [g.nnz for g in graphs]
        torch.manual_seed(0)
        model = ctgcn_amd.CTGCN(10, 12, 8, 1, 2, T, rnn_type="GRU", model_type="S", trans_activate_type="N")
        x_all = [torch.from_numpy(a) for a in formula_tensor((T, n, 10), 0.21, 0.4)]
        gsel = torch.from_numpy(formula_tensor((T, n, 8), 0.37, 1.1))

        # unsharded truth from the CPU oracle, grads via autograd on cloned leaf weights
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        ref_out, ref_trans = TP.ctgcn_with_grad(sd, x_all, [[TP.coo_like_reference(m) for m in l] for l in lists], "GRU", "S", "N")
        (ref_out * gsel).sum().backward()

        plan = spp.shard_ctgcn(model, n, costs=costs, exchange=exchange, gather_output=False)
        mine = plan.assignment[rank]
        assert sorted(sum(plan.assignment, [])) == list(range(T))
        x_list = [x_all[t] if t in mine else None for t in range(T)]
        adj_list = [CoreAdj.from_matrices(lists[t]) if t in mine else None for t in range(T)]
        out, trans = model(x_list, adj_list)
        lo, hi = plan.node_range(rank)
        assert out.shape == (T, hi - lo, 8)
        err_fwd = (out - ref_out[:, lo:hi]).abs().max().item()
        for t in mine:
            assert torch.allclose(trans[t], ref_trans[t], atol=1e-6)
        (out * gsel[:, lo:hi]).sum().backward()
        spp.allreduce_replicated_grads(model)
        err_bwd, compared = 0.0, 0
        owned = {id(p) for p in spp.owned_parameters(model)}
        for name, p in model.named_parameters():
            ref_g = sd[name].grad
            if id(p) not in owned or ref_g is None:      # CoreDiffusion.linear is unused -> no grad on either side
                continue
            assert p.grad is not None, name
            err_bwd = max(err_bwd, (p.grad - ref_g).abs().max().item() / (1e-6 + ref_g.abs().max().item()))
            compared += 1
        assert compared >= 10 * len(mine) + 6, compared
        # full-output mode returns the reference's [T, N, d] on every rank
        model.shard_gather_output = True
        with torch.no_grad():
            full, _ = model(x_list, adj_list)
            again, _ = model(x_list, adj_list)                 # second inference forward: the cached exchange buffers are reused
        assert torch.equal(full, again)
        err_full = (full - ref_out).abs().max().item()
        # training on the gathered output (replicated-loss convention: every rank computes the SAME loss on the full
        # [T, N, d]): parameter gradients must equal the reference's, not world x them
        model.zero_grad()
        full, _ = model(x_list, adj_list)
        (full * gsel).sum().backward()
        spp.allreduce_replicated_grads(model)
        err_bwd_full = 0.0
        for name, p in model.named_parameters():
            ref_g = sd[name].grad
            if id(p) not in owned or ref_g is None:
                continue
            err_bwd_full = max(err_bwd_full, (p.grad - ref_g).abs().max().item() / (1e-6 + ref_g.abs().max().item()))
        results[rank] = (err_fwd, max(err_bwd, err_bwd_full), err_full, plan.assignment)
    finally:
        dist.destroy_process_group()


def run_cgcn(rank, world, port, T, n, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctgcn_amd
        from ctgcn_amd import CoreAdj, ops, snapshot_parallel as spp
        from ctgcn_amd.synth import dynamic_graph
        from oracle import oracle as O, torch_path as TP
        from conftest import formula_tensor
        ops.core_aggregate = _oracle_aggregate
        graphs = dynamic_graph(n, 6, T, seed=5)
        lists = [O.core_adj_list([O.kcore_matrices(g)], 0, 1, 1, max_core=3)[0] for g in graphs]
        torch.manual_seed(100 + rank)                       # replicas start DIFFERENT: shard_cgcn must broadcast rank 0's
        model = ctgcn_amd.CGCN(10, 12, 8, 2, 2, rnn_type="GRU", model_type="C", trans_activate_type="N")
        assignment = spp.shard_cgcn(model, T, costs=[g.nnz for g in graphs])
        x_all = [torch.from_numpy(a) for a in formula_tensor((T, n, 10), 0.21, 0.4)]
        gsel = torch.from_numpy(formula_tensor((T, n, 8), 0.37, 1.1))
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        ref = TP.cgcn_with_grad(sd, x_all, [[TP.coo_like_reference(m) for m in l] for l in lists], "GRU", "C", "N")
        sum((r * gsel[t]).sum() for t, r in enumerate(ref)).backward()
        mine = assignment[rank]
        adj = [CoreAdj.from_matrices(lists[t]) if t in mine else None for t in range(T)]
        out = model([x_all[t] if t in mine else None for t in range(T)], adj)
        err_fwd = max((out[t] - ref[t]).abs().max().item() for t in mine)
        assert all(out[t] is None for t in range(T) if t not in mine)
        sum((out[t] * gsel[t]).sum() for t in mine).backward()
        spp.allreduce_grads(model)
        err_bwd = 0.0
        for name, p in model.named_parameters():
            g = sd[name].grad
            if g is None:
                continue
            err_bwd = max(err_bwd, (p.grad - g).abs().max().item() / (1e-6 + g.abs().max().item()))
        results[rank] = (err_fwd, err_bwd, assignment)
    finally:
        dist.destroy_process_group()


def run_shared_seed(rank, world, port, results):
    """snapshot_parallel.share_loss_seed: every rank ends up with rank 0's entropy-drawn stream base (and a reset call counter)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ctgcn_amd import snapshot_parallel as spp

        class Loss(object):
            seed = None
            shared_base = None
            _shared_calls = 7
        loss = Loss()
        base = spp.share_loss_seed(loss)
        results[rank] = (base, loss.shared_base, loss._shared_calls)
    finally:
        dist.destroy_process_group()
