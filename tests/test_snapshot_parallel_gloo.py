"""N>1 path on CPU: world_size 2 and 4, gloo.  Covers plan_assignment, both exchange modes, uneven T/world (incl. a rank that owns
nothing), the pipelined inference exchange with its cached buffers, autograd through the collectives and the replicated-head grad
all-reduce."""
import socket

import pytest
import torch.multiprocessing as mp

from ctgcn_amd import snapshot_parallel as spp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_plan_assignment_balances_cumulative_snapshots():
    costs = [i + 1 for i in range(16)]
    plan = spp.plan_assignment(costs, 8)
    loads = [sum(costs[t] for t in r) for r in plan]
    assert sorted(sum(plan, [])) == list(range(16)) and all(len(r) == 2 for r in plan)
    assert max(loads) == min(loads) == 17            # pairs (t, 15-t); round-robin would give 10..24
    assert spp.plan_assignment([5, 1, 1], 2) == [[0], [1, 2]]
    assert spp.plan_assignment([1.0] * 3, 4) == [[0], [1], [2], []]
    p = spp.ShardPlan([[0, 3], [1, 2]], 10)
    assert (p.per, p.n_slice, p.n_pad) == (2, 5, 10) and p.owner(3) == 0 and p.node_range(1) == (5, 10)


@pytest.mark.parametrize("exchange,T,n,world", [("all_to_all", 5, 301, 2), ("all_gather", 4, 300, 2),
                                                 ("all_to_all", 5, 203, 4),      # uneven: one rank owns two snapshots, three own one
                                                 ("all_to_all", 3, 150, 4),      # a rank that owns nothing still takes part in the exchange
                                                 ("all_gather", 6, 150, 4),
                                                 ("all_to_all", 16, 96, 8)])     # the north-star layout: 16 cumulative snapshots on 8 ranks, two each (LPT)
def test_sharded_ctgcn_matches_unsharded(exchange, T, n, world):
    from _dist_worker import run
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(run, args=(world, _free_port(), T, n, exchange, results), nprocs=world, join=True)
    assert len(results) == world
    if T < world:
        assert [] in results[0][3]
    if T == 2 * world:
        assert all(len(a) == 2 for a in results[0][3])       # cumulative snapshots grow with t: LPT gives every rank two, a small one with a large one
        loads = [sum(t + 1 for t in a) for a in results[0][3]]
        assert max(loads) - min(loads) <= 2, results[0][3]
    for rank in range(world):
        err_fwd, err_bwd, err_full, assignment = results[rank]
        assert err_fwd < 1e-5 and err_full < 1e-5, (rank, err_fwd, err_full)
        assert err_bwd < 1e-4, (rank, err_bwd)


def test_sharded_cgcn_is_data_parallel():
    from _dist_worker import run_cgcn
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(run_cgcn, args=(2, _free_port(), 5, 200, results), nprocs=2, join=True)
    for rank in range(2):
        err_fwd, err_bwd, assignment = results[rank]
        assert err_fwd < 1e-5 and err_bwd < 1e-4, (rank, err_fwd, err_bwd)
        assert sorted(sum(assignment, [])) == list(range(5))


def test_share_loss_seed_gives_every_rank_the_same_stream():
    from _dist_worker import run_shared_seed
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(run_shared_seed, args=(3, _free_port(), results), nprocs=3, join=True)
    bases = {results[r][0] for r in range(3)}
    assert len(bases) == 1 and all(results[r][1] == results[r][0] and results[r][2] == 0 for r in range(3))
    assert next(iter(bases)) > 0
