"""Worker for tests/test_gpu_nccl.py: one process per GPU under RCCL (backend "nccl"), world size 1 or 2.
Checks the snapshot-parallel forward (both exchange modes, pipelined inference path) and backward
(+ allreduce_replicated_grads) against the unsharded HIP forward/backward of the same model on the same GPU."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(rank, world, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import ctgcn_amd
        from ctgcn_amd import snapshot_parallel as spp
        from ctgcn_amd.helper import core_adj_from_scipy
        from ctgcn_amd.synth import dynamic_graph
        n, T = 3001, 5
        graphs = dynamic_graph(n, 8, T, seed=9)
        adj = [core_adj_from_scipy(g, 4, dev)[0] for g in graphs]
        torch.manual_seed(0)
        model = ctgcn_amd.CTGCN(20, 128, 128, 1, 2, T).to(dev)
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
        torch.manual_seed(1)
        xs = [torch.randn(n, 20, device=dev) for _ in range(T)]
        gsel = torch.randn(T, n, 128, device=dev)

        # unsharded truth on this GPU
        with torch.no_grad():
            want = model(xs, adj).clone()
        model.zero_grad()
        out = model(xs, adj)
        assert torch.equal(out.detach(), want)
        (out * gsel).sum().backward()
        want_grad = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

        report = {}
        for exchange in ("all_to_all", "all_gather"):
            model.process_group = None
            plan = spp.shard_ctgcn(model, n, costs=[g.nnz for g in graphs], exchange=exchange, gather_output=False)
            mine = plan.assignment[rank]
            lo, hi = plan.node_range(rank)
            x_l = [xs[t] if t in mine else None for t in range(T)]
            a_l = [adj[t] if t in mine else None for t in range(T)]
            with torch.no_grad():                       # inference: the pipelined path for all_to_all
                got = model(x_l, a_l)
            report[exchange + "_fwd_bitwise"] = bool(torch.equal(got, want[:, lo:hi]))
            report[exchange + "_fwd_err"] = float((got - want[:, lo:hi]).abs().max())
            model.zero_grad()
            got = model(x_l, a_l)                       # autograd path
            report[exchange + "_train_fwd_bitwise"] = bool(torch.equal(got.detach(), want[:, lo:hi]))
            (got * gsel[:, lo:hi]).sum().backward()
            spp.allreduce_replicated_grads(model)
            owned = {id(p) for p in spp.owned_parameters(model)}
            worst = 0.0
            for k, p in model.named_parameters():
                if id(p) not in owned or k not in want_grad:
                    continue
                assert p.grad is not None, k
                worst = max(worst, float((p.grad - want_grad[k]).abs().max()) / (1e-12 + float(want_grad[k].abs().max())))
            report[exchange + "_grad_rel_err"] = worst
            # full-output mode
            model.shard_gather_output = True
            with torch.no_grad():
                full = model(x_l, a_l)
            report[exchange + "_gather_bitwise"] = bool(torch.equal(full, want))
        model.process_group = None
        results[rank] = report
    finally:
        dist.destroy_process_group()
