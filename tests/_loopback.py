"""G "virtual ranks" of the snapshot-parallel path in ONE process on ONE GPU (tests/test_gpu_virtual_ranks.py).

The GPU box has one MI355X, so RCCL never runs with more than one rank there, and the gloo tests run the CPU branches with the
oracle injected.  What neither covers is the GPU-side geometry of world > 1: the send buffer written by the last CoreDiffusion of a
snapshot, the receive buffer `recv [per, world, n_slice, d]`, the per-step offset table the temporal GRU kernel reads it through,
node slices with padding, ranks that own nothing.  Here every virtual rank is a Python thread with its own copy of the model;
`ctgcn_amd.snapshot_parallel.dist` is replaced by `LoopbackDist`, whose collectives are device copies between the ranks' buffers.

Scheduling is cooperative, not concurrent: a baton (lock) lets exactly one rank run Python / enqueue GPU work at a time, and a rank
hands it over only inside a collective.  `ctgcn_amd.ops` keeps process-wide state (operand-plane cache, side streams), which real
deployments never share between ranks (one process per GPU); the baton keeps that true here.  Every collective fences the device on
both sides, so the copies see finished producers whatever stream they ran on.
"""
import threading

import torch


class _Work(object):
    def wait(self):
        return True

    def is_completed(self):
        return True


class _Group(object):
    def __init__(self):
        self.WORLD = "loopback-world"


class _ReduceOp(object):
    SUM = "sum"
    MAX = "max"


class LoopbackDist(object):
    """The subset of torch.distributed that ctgcn_amd.snapshot_parallel uses, for `world` threads of this process."""

    def __init__(self, world, backend="nccl", timeout=300.0):
        self.world = world
        self.backend = backend
        self.timeout = timeout
        self.group = _Group()
        self.ReduceOp = _ReduceOp
        self._tls = threading.local()
        self._baton = threading.Lock()
        self._barrier = threading.Barrier(world)
        self._box = [None] * world
        self.calls = {}                       # collective name -> number of calls (rank 0's count)

    # ----------------------------------------------------------------------------- running the ranks
    def run(self, fn):
        """fn(rank) on every virtual rank; returns the list of results; the first exception of any rank is re-raised."""
        results, errors = [None] * self.world, [None] * self.world

        def body(r):
            self._tls.rank = r
            self._baton.acquire()
            try:
                # backward nodes on the CALLING thread: the autograd engine's per-device worker thread is shared by every thread of the
                # process, and a collective blocking inside it would wait for ranks whose backward is queued behind it
                with torch.autograd.set_multithreading_enabled(False):
                    results[r] = fn(r)
            except BaseException as exc:          # noqa: B902 - reported below, with the other ranks released
                errors[r] = exc
                self._barrier.abort()
            finally:
                self._baton.release()
        threads = [threading.Thread(target=body, args=(r,), name="vrank%d" % r) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        real = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real:
            raise real[0]
        if any(errors):
            raise [e for e in errors if e is not None][0]
        return results

    def _meet(self):
        """Every rank arrives, with its GPU work finished; the baton is free while waiting."""
        torch.cuda.synchronize()
        self._baton.release()
        try:
            self._barrier.wait(self.timeout)
        finally:
            self._baton.acquire()

    def _count(self, name):
        if self.get_rank() == 0:
            self.calls[name] = self.calls.get(name, 0) + 1

    # ----------------------------------------------------------------------------- torch.distributed's face
    def get_rank(self, group=None):
        return self._tls.rank

    def get_world_size(self, group=None):
        return self.world

    def get_backend(self, group=None):
        return self.backend

    def get_global_rank(self, group, r):
        return r

    def barrier(self, group=None):
        self._meet()

    def all_to_all_single(self, output, input, group=None, async_op=False):
        self._count("all_to_all_single")
        r, w = self.get_rank(), self.world
        assert output.is_contiguous() and input.is_contiguous() and output.numel() == input.numel() and input.numel() % w == 0
        self._box[r] = input
        self._meet()
        out = output.view(w, -1)
        for src in range(w):
            out[src].copy_(self._box[src].view(w, -1)[r])
        self._meet()                                  # nobody reuses its send buffer before every reader is done
        return _Work() if async_op else None

    def all_gather_into_tensor(self, output, input, group=None, async_op=False):
        self._count("all_gather_into_tensor")
        r, w = self.get_rank(), self.world
        assert output.is_contiguous() and input.is_contiguous() and output.numel() == w * input.numel()
        self._box[r] = input
        self._meet()
        out = output.view(w, -1)
        for src in range(w):
            out[src].copy_(self._box[src].reshape(-1))
        self._meet()
        return _Work() if async_op else None

    def reduce_scatter_tensor(self, output, input, group=None, async_op=False):
        self._count("reduce_scatter_tensor")
        r, w = self.get_rank(), self.world
        assert input.is_contiguous() and input.numel() == w * output.numel()
        self._box[r] = input
        self._meet()
        acc = self._box[0].view(w, -1)[r].clone()
        for src in range(1, w):
            acc += self._box[src].view(w, -1)[r]
        self._meet()
        output.copy_(acc.view_as(output))
        return _Work() if async_op else None

    def all_reduce(self, tensor, op=None, group=None, async_op=False):
        self._count("all_reduce")
        r, w = self.get_rank(), self.world
        self._box[r] = tensor.clone()
        self._meet()
        if op == _ReduceOp.MAX:
            acc = self._box[0].clone()
            for src in range(1, w):
                acc = torch.maximum(acc, self._box[src])
        else:
            acc = self._box[0].clone()
            for src in range(1, w):
                acc += self._box[src]
        self._meet()
        tensor.copy_(acc)
        return _Work() if async_op else None

    def broadcast(self, tensor, src=0, group=None, async_op=False):
        self._count("broadcast")
        if self.get_rank() == src:
            self._box[src] = tensor
        self._meet()
        if self.get_rank() != src:
            tensor.copy_(self._box[src])
        self._meet()
        return _Work() if async_op else None
