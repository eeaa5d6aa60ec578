"""Snapshot-parallel CTGCN: one process per GPU, snapshots of a temporal window sharded over ranks.

Until the temporal RNN (reference models.py:248) every snapshot t is independent AND owns its own weights
(mlp_list[t], duffision_list[t], models.py:225-231), so a rank that owns snapshot t needs only that
snapshot's graph, features, weights and optimizer state.  There is exactly one exchange step, placed where
the reference stacks the per-snapshot states (models.py:248):

  exchange="all_to_all" (default)  each rank keeps ONLY its node slice [N/G] of every snapshot's state and
      runs the temporal GRU/LSTM + LayerNorm on that slice: one all-to-all of (T/G)·N·d·4 B sent per rank
      (1/G of an all-gather's received volume), and the temporal step scales with G as well.
  exchange="all_gather"            every rank receives every snapshot's full state (RCCL all-gather over
      xGMI, as BASELINE.json's north_star words it) and runs the temporal step on its node slice
      (or on all nodes with replicate_head=True).

Backward is the transposed collective (all-to-all / reduce-scatter).  The temporal rnn/norm weights are
replicated; sum their grads with allreduce_replicated_grads() after backward.  With gather_output=True the
output node slices are all-gathered so every rank returns the reference's full [T, N, d]; training on that output
follows the replicated-loss convention — every rank computes the SAME loss on the full tensor (as the reference's
single process does) and the gather's backward keeps the rank's own block of the gradient, so parameter gradients equal
the reference's.  With gather_output=False (bench.py, large graphs) the loss is computed on the rank's node slice.

The backend is whatever the process group was created with: "nccl" (= RCCL) on MI355X, "gloo" in CPU tests.
"""
import math

import torch
import torch.distributed as dist


# --------------------------------------------------------------------------------------- planning
def plan_assignment(costs, world):
    """Greedy longest-processing-time assignment of snapshots to ranks, at most ceil(T/world) each.
    costs[t]: relative cost of snapshot t (e.g. aggregated edges).  Returns list[world] of sorted index lists.
    Cumulative dynamic graphs grow with t (reference graph.py:101-108), so round-robin would leave the last
    rank ~40% over the mean at T=16, G=8."""
    T = len(costs)
    cap = math.ceil(T / world) if T else 0
    load = [0.0] * world
    owned = [[] for _ in range(world)]
    for t in sorted(range(T), key=lambda i: (-float(costs[i]), i)):
        r = min((r for r in range(world) if len(owned[r]) < cap), key=lambda r: (load[r], r))
        owned[r].append(t)
        load[r] += float(costs[t])
    return [sorted(o) for o in owned]


class ShardPlan(object):
    def __init__(self, assignment, num_nodes):
        self.assignment = [list(a) for a in assignment]
        self.world = len(assignment)
        self.T = sum(len(a) for a in assignment)
        self.per = max((len(a) for a in assignment), default=0)       # slots per rank in the exchange buffer
        self.n = int(num_nodes)
        self.n_slice = (self.n + self.world - 1) // self.world
        self.n_pad = self.n_slice * self.world
        # position of snapshot t in the rank-major [world, per] exchange layout
        self.slot_of = {}
        for r, lst in enumerate(self.assignment):
            for i, t in enumerate(lst):
                self.slot_of[t] = r * self.per + i
        assert sorted(self.slot_of) == list(range(self.T)), "assignment must cover 0..T-1 exactly once"

    def owner(self, t):
        return self.slot_of[t] // self.per

    def node_range(self, rank):
        lo = rank * self.n_slice
        return lo, min(self.n, lo + self.n_slice)


def shard_ctgcn(model, num_nodes, costs=None, assignment=None, group=None, exchange="all_to_all", gather_output=True,
                replicate_head=False):
    """Turn a CTGCN into its snapshot-parallel form on the current process group. Returns the ShardPlan."""
    assert exchange in ("all_to_all", "all_gather")
    group = group if group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    if assignment is None:
        costs = [1.0] * model.duration if costs is None else costs
        assignment = plan_assignment(costs, world)
    plan = ShardPlan(assignment, num_nodes)
    assert plan.T == model.duration and plan.world == world
    release_exchange_buffers(model)                  # buffers of an earlier plan
    model.process_group, model.shard_plan = group, plan
    model.shard_exchange, model.shard_gather_output, model.shard_replicate_head = exchange, gather_output, replicate_head
    return plan


def share_loss_seed(loss, group=None, src=0):
    """Give every rank's NegativeSamplingLoss(seed=None) the same draw stream (see the module docstring: the replicated-loss convention
    of gather_output=True needs identical samples on every rank): rank `src` draws a base from OS entropy and broadcasts it."""
    import os
    group = group if group is not None else dist.group.WORLD
    base = [int.from_bytes(os.urandom(7), "little")]
    dist.broadcast_object_list(base, src=src, group=group)
    loss.shared_base = int(base[0])
    loss._shared_calls = 0
    return loss.shared_base


def owned_parameters(model):
    """Parameters this rank must optimise: its snapshots' mlp/CDN weights + the replicated temporal head."""
    rank = dist.get_rank(model.process_group)
    mine = model.shard_plan.assignment[rank]
    for t in mine:
        yield from model.mlp_list[t].parameters()
        yield from model.duffision_list[t].parameters()
    yield from model.rnn.parameters()
    yield from model.norm.parameters()


def allreduce_replicated_grads(model):
    """Sum the grads of the replicated temporal rnn/norm weights (each rank saw only its node slice)."""
    if model.shard_replicate_head:
        return
    for p in list(model.rnn.parameters()) + list(model.norm.parameters()):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        dist.all_reduce(p.grad, group=model.process_group)


# ------------------------------------------------------------------------ collectives with autograd
class _TimeToNodeAllToAll(torch.autograd.Function):
    """[per, n_pad, d] (my snapshots, all nodes) -> [world, per, n_slice, d] (all snapshots, my nodes)."""

    @staticmethod
    def forward(ctx, local, group):
        world = dist.get_world_size(group)
        per, n_pad, d = local.shape
        ctx.group, ctx.shape = group, (per, n_pad, d)
        send = local.view(per, world, n_pad // world, d).transpose(0, 1).contiguous()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=group)
        return recv

    @staticmethod
    def backward(ctx, grad):
        per, n_pad, d = ctx.shape
        grad = grad.contiguous()
        back = torch.empty_like(grad)
        dist.all_to_all_single(back, grad, group=ctx.group)
        return back.transpose(0, 1).reshape(per, n_pad, d), None


class _AllGather(torch.autograd.Function):
    """x [..] -> [world, ..]; backward = reduce-scatter (sum of every rank's grad for my contribution)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        # concatenated-along-dim-0 output layout: the one every backend (RCCL and gloo) accepts
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out.view((world,) + tuple(x.shape))

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        out = torch.empty(grad.shape[1:], dtype=grad.dtype, device=grad.device)
        if dist.get_backend(ctx.group) == "gloo":      # gloo has no reduce_scatter
            grad = grad.clone()                            # autograd may hand the same buffer to other consumers
            dist.all_reduce(grad, group=ctx.group)
            out.copy_(grad[dist.get_rank(ctx.group)])
        else:
            dist.reduce_scatter_tensor(out, grad.view((-1,) + tuple(grad.shape[2:])), group=ctx.group)
        return out, None


class _GatherReplicatedOutput(torch.autograd.Function):
    """x [..] -> [world, ..] for an output that every rank then feeds to the SAME loss (gather_output=True: each rank
    returns the reference's full [T, N, d]).  Every rank's incoming gradient is then the same full tensor, so the
    gradient of this rank's contribution is simply its own block of it — no communication, and no world-fold
    over-count as a reduce-scatter of identical gradients would give."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out.view((world,) + tuple(x.shape))

    @staticmethod
    def backward(ctx, grad):
        return grad[dist.get_rank(ctx.group)].contiguous(), None


# ------------------------------------------------------------------------------------ sharded forward
def ctgcn_forward_sharded(model, x_list, adj_list):
    """CTGCN.forward under a ShardPlan.  x_list / adj_list have length T; entries of snapshots this rank does
    not own are ignored (may be None)."""
    group, plan = model.process_group, model.shard_plan
    rank = dist.get_rank(group)
    mine = plan.assignment[rank]
    assert len(x_list) == plan.T, "window length %d != plan %d" % (len(x_list), plan.T)

    needs_grad = model._needs_autograd(x_list)
    if model.shard_exchange == "all_to_all" and not needs_grad and plan.per > 0:
        return _forward_sharded_pipelined(model, x_list, adj_list)

    states, trans_local = [], {}
    for t in mine:
        h, tr = model.snapshot_branch(t, x_list[t], adj_list[t])
        states.append(h)
        trans_local[t] = tr
    ref = states[0] if states else None
    if ref is None:      # a rank may own nothing when T < world: it still takes part in the exchange
        p = next(model.rnn.parameters())
        ref = torch.zeros(plan.n, model.output_dim, dtype=p.dtype, device=p.device)
    d = ref.shape[1]
    pad_rows = plan.n_pad - plan.n
    rows = [torch.nn.functional.pad(h, (0, 0, 0, pad_rows)) if pad_rows else h for h in states]
    while len(rows) < plan.per:                      # uneven T/world: empty slot, never read back
        rows.append(torch.zeros(plan.n_pad, d, dtype=ref.dtype, device=ref.device))
    local = torch.stack(rows)                        # [per, n_pad, d]
    if needs_grad and not local.requires_grad:
        # a rank that owns no snapshot (T < world) has nothing upstream of the exchange, but the transposed collective in backward is
        # collective too: without a graph node here this rank would skip it and the others would wait for ever
        local.requires_grad_(True)

    order = [plan.slot_of[t] for t in range(plan.T)]
    lo, hi = plan.node_range(rank)
    if model.shard_exchange == "all_to_all":
        got = _TimeToNodeAllToAll.apply(local, group)                       # [world, per, n_slice, d]
        seq = got.reshape(plan.world * plan.per, plan.n_slice, d)[order]    # time order, my nodes
        seq = seq[:, : hi - lo]
    else:
        got = _AllGather.apply(local, group)                                # [world, per, n_pad, d]
        seq = got.reshape(plan.world * plan.per, plan.n_pad, d)[order]
        seq = seq[:, : plan.n] if model.shard_replicate_head else seq[:, lo:hi]

    out = model.temporal_head(seq.transpose(0, 1))                          # [T, nodes, d]
    if not model.shard_replicate_head and model.shard_gather_output:
        padded = torch.nn.functional.pad(out, (0, 0, 0, plan.n_slice - (hi - lo))) if hi - lo < plan.n_slice else out
        full = _GatherReplicatedOutput.apply(padded.transpose(0, 1).contiguous(), group)  # [world, n_slice, T, d]
        out = full.reshape(plan.n_pad, plan.T, d)[: plan.n].transpose(0, 1)
    if model.model_type == 'C':
        return out
    return out, [trans_local.get(t) for t in range(plan.T)]


def release_exchange_buffers(model):
    """Drop the send / receive / gather buffers the pipelined inference exchange keeps on the model ((T/G) N d floats, two or three
    times over).  They are re-allocated by the next sharded inference forward; call this when the model goes back to training or
    the plan changes and the memory is wanted."""
    if getattr(model, "_exchange_buffers", None) is not None:
        model._exchange_buffers = None


def _slot_moves(plan, device):
    """per exchange slot s: (time indices of the ranks' s-th snapshots, the ranks that have one (None = the first `count`), count)"""
    moves = []
    for s_ in range(plan.per):
        owners = [w_ for w_ in range(plan.world) if s_ < len(plan.assignment[w_])]
        if not owners:
            moves.append(None)
            continue
        times = torch.tensor([plan.assignment[w_][s_] for w_ in owners], device=device)
        leading = owners == list(range(len(owners)))
        moves.append((times, None if leading else torch.tensor(owners, device=device), len(owners)))
    return moves


def _forward_sharded_pipelined(model, x_list, adj_list):
    """Inference form of the all-to-all exchange, overlapped with compute: slot s of every rank (its s-th snapshot) is
    exchanged by its own asynchronous all-to-all as soon as it is computed, on the collective's stream, while the next
    snapshot runs on the compute stream.  The last CoreDiffusion of a snapshot writes its embeddings straight into the send
    buffer.  At G = 2 the whole exchange is N/2 x T/2 x d x 4 B over ONE xGMI link (2 GB for config 5, ~40 ms if done at the
    end); per slot it is 1/per of that and hides behind a snapshot's ~16 ms of compute.  Same numbers as the autograd path."""
    group, plan = model.process_group, model.shard_plan
    rank = dist.get_rank(group)
    mine = plan.assignment[rank]
    p0 = next(model.rnn.parameters())
    d = model.output_dim
    lo, hi = plan.node_range(rank)
    # exchange buffers live on the model: allocating and zeroing (T/G) N d floats three times per forward cost the single-rank RCCL run
    # 27 of 232 ms (profiles/r02_bench_1gpu_forced_dist.json).  Pad rows and empty slots of `send` are zeroed once and never written.
    key = (plan.per, plan.n_pad, plan.world, plan.n_slice, plan.T, hi - lo, d, p0.dtype, p0.device)
    buf = getattr(model, "_exchange_buffers", None)
    if buf is None or buf[0] != key:
        buf = model._exchange_buffers = (key,
                                         torch.zeros(plan.per, plan.n_pad, d, dtype=p0.dtype, device=p0.device),
                                         torch.empty(plan.per, plan.world, plan.n_slice, d, dtype=p0.dtype, device=p0.device),
                                         [None],                          # [nodes, T, d] gather buffer: only when the temporal step cannot read recv itself
                                         _slot_moves(plan, p0.device),
                                         # element offset of snapshot t's block inside recv [per, world, n_slice, d]
                                         torch.tensor([((plan.slot_of[t] % plan.per) * plan.world + plan.slot_of[t] // plan.per) * plan.n_slice * d
                                                       for t in range(plan.T)], dtype=torch.int64, device=p0.device))
    _, send, recv, seq_box, moves, step_off = buf
    works, trans_local = [], {}
    # model.shard_timing = [] (bench.py --gpus N): per forward, HIP events at the phase boundaries on the compute stream — snapshot
    # branches | waiting for the exchange (what is NOT hidden behind compute) | temporal head; read with shard_phase_ms()
    marks = getattr(model, "shard_timing", None)

    def mark():
        if marks is not None and send.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            return ev
        return None
    ev0 = mark()
    for s_ in range(plan.per):
        if s_ < len(mine):
            t = mine[s_]
            h, tr = model.snapshot_branch(t, x_list[t], adj_list[t], out=send[s_, : plan.n])
            if h.data_ptr() != send[s_].data_ptr():
                send[s_, : plan.n].copy_(h)
            trans_local[t] = tr
        works.append(dist.all_to_all_single(recv[s_].view(plan.world * plan.n_slice, d), send[s_], group=group, async_op=True))
    # recv[s][w] = snapshot assignment[w][s] on my node slice -> [nodes, T, d] in time order (the temporal GRU's input layout); slot s is
    # moved as soon as ITS all-to-all has landed (one strided copy per slot: the ranks' s-th snapshots), under the later slots' exchange
    from . import ops
    ev1 = mark()
    ev2 = None
    if recv.is_cuda and d == 128 and ops.gru_steps_scattered_ok(model.rnn, recv):
        # the temporal GRU reads the receive buffer in time order through a per-step offset table: no [nodes, T, d] copy at all
        for w in works:
            w.wait()
        ev2 = mark()
        out = ops.gru_sequence_scattered(model.rnn, model.norm, recv, step_off, d, hi - lo).transpose(0, 1)        # [T, my nodes, d]
    else:
        if seq_box[0] is None:
            seq_box[0] = torch.empty(hi - lo, plan.T, d, dtype=p0.dtype, device=p0.device)
        seq = seq_box[0]
        for s_ in range(plan.per):
            works[s_].wait()
            if moves[s_] is not None:
                times, owners, count = moves[s_]
                src = recv[s_, :count, : hi - lo] if owners is None else recv[s_, owners, : hi - lo]
                seq.index_copy_(1, times, src.transpose(0, 1))
        ev2 = mark()
        out = model.temporal_head(seq)                                                      # [T, my nodes, d]
    ev3 = mark()
    if marks is not None and ev0 is not None:
        marks.append((ev0, ev1, ev2, ev3))
    if model.shard_gather_output:
        padded = torch.nn.functional.pad(out, (0, 0, 0, plan.n_slice - (hi - lo))) if hi - lo < plan.n_slice else out
        full = _GatherReplicatedOutput.apply(padded.transpose(0, 1).contiguous(), group)    # [world, n_slice, T, d]
        out = full.reshape(plan.n_pad, plan.T, d)[: plan.n].transpose(0, 1)
    if model.model_type == 'C':
        return out
    return out, [trans_local.get(t) for t in range(plan.T)]


def shard_phase_ms(model, last=None):
    """Mean (snapshot branches, exposed exchange wait, temporal head) milliseconds of this rank over the recorded forwards of
    model.shard_timing (the last `last` of them), or None.  Call after a device synchronize."""
    marks = getattr(model, "shard_timing", None)
    if not marks:
        return None
    use = marks[-last:] if last else marks
    k = float(len(use))
    return {"snapshot_branches_ms": sum(a.elapsed_time(b) for a, b, _, _ in use) / k,
            "exchange_exposed_ms": sum(b.elapsed_time(c) for _, b, c, _ in use) / k,
            "temporal_head_ms": sum(c.elapsed_time(d_) for _, _, c, d_ in use) / k, "forwards": len(use)}


# ------------------------------------------------------------------------------------------- CGCN (static model)
def shard_cgcn(model, num_snapshots, costs=None, assignment=None, group=None):
    """CGCN shares ONE set of weights across snapshots (reference models.py:158-163), so the window is plain
    data parallelism: every rank keeps a full replica, embeds only its snapshots, and weight gradients are summed
    with allreduce_grads() after backward.  No exchange step is needed in the forward.  Returns the assignment."""
    group = group if group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    if assignment is None:
        assignment = plan_assignment([1.0] * num_snapshots if costs is None else costs, world)
    model.process_group, model.shard_assignment = group, [list(a) for a in assignment]
    src = dist.get_global_rank(group, 0) if group is not dist.group.WORLD else 0
    with torch.no_grad():                              # replicas must start identical; the parameter itself (not .data): an in-place
        for p in model.parameters():                   # write the version counter sees, so cached operand planes of a weight re-split
            dist.broadcast(p, src=src, group=group)
    from . import ops
    ops.invalidate_plane_cache()
    return model.shard_assignment


def cgcn_forward_sharded(model, x_list, adj_list):
    """Embeds the snapshots this rank owns; entries of other snapshots are None in the returned list(s)."""
    rank = dist.get_rank(model.process_group)
    mine = set(model.shard_assignment[rank])
    emb, trans = [None] * len(x_list), [None] * len(x_list)
    for t in sorted(mine):
        res = model.cgcn(x_list[t], adj_list[t])
        emb[t], trans[t] = (res if model.model_type == 'S' else (res, None))
    return emb if model.model_type == 'C' else (emb, trans)


def allreduce_grads(model):
    """Sum every parameter gradient over the ranks (data-parallel CGCN)."""
    for p in model.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        dist.all_reduce(p.grad, group=model.process_group)
