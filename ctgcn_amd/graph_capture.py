"""HIP-graph capture of a whole CTGCN / CGCN inference window.

On small graphs (BASELINE configs 2-4: thousands of nodes) a window is ~300 kernel launches of a few microseconds each
and the step time is launch + Python overhead, not GPU time.  The whole forward — aggregation, projection, recurrence,
LayerNorm, temporal GRU — only launches kernels on the current stream and allocates through torch's caching allocator
(no host synchronisation once the adjacency caches are warm), so it can be recorded ONCE into a hipGraph and replayed
with a single launch.  Adjacency structure and weights are baked in by address: replay sees in-place weight updates
(`load_state_dict`, optimizer steps) but a different graph needs a new capture.  That contract has a price: the operand forms the
eager path caches per weight version (packed W planes, folded GRU biases, the planes of static features) are rebuilt INSIDE the
graph on every replay.  `frozen_weights=True` records the cached forms instead — the replay is then pure layer kernels — and the
caller re-captures after changing a weight or a static feature tensor (an embedding run never does, reference embedding.py:302-318).

    runner = GraphedInference(model, x_list, adj_list)     # warm-up + capture
    emb = runner()                                          # replay; same tensors as model(x_list, adj_list)
    emb = runner(new_x_list)                                # dense features are copied into the captured input buffers

Reference call being replaced: embedding.py:302,318 `model(x_list, adj_list)` under torch.no_grad().
"""
import torch


def _is_dense(x):
    return torch.is_tensor(x) and not x.is_sparse


class GraphedInference:
    def __init__(self, model, x_list, adj_list, warmup=2, frozen_weights=False):
        single = not isinstance(x_list, (list, tuple))
        xs = [x_list] if single else list(x_list)
        if not all(x is None or torch.is_tensor(x) for x in xs):
            raise TypeError("GraphedInference: features must be tensors")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedInference needs the model on an MI355X (cuda device), got %s" % dev)
        if model.training:
            raise RuntimeError("GraphedInference captures inference: call model.eval() first")
        # dense features get private input buffers (replay reads these addresses); sparse one-hot features carry no data
        # (frozen_weights: a feature tensor the caller has declared static — ops.mark_static, what the loader returns — is read in place, so
        # that its cached operand planes are the ones recorded; replaying with other features needs a new capture for those)
        from . import ops as _ops
        self._static_x = [x if (frozen_weights and _is_dense(x) and _ops.is_static(x)) else (x.detach().clone() if _is_dense(x) else x) for x in xs]
        self._single = single
        self._model, self._adj = model, adj_list
        arg = self._static_x[0] if single else self._static_x
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):      # fills the CoreAdj / identity / hub-row caches (they synchronise once)
                model(arg, adj_list)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        self.frozen_weights = bool(frozen_weights)
        self._held = []                          # frozen: the cached operand buffers the graph reads (kept alive with it)
        from . import ops
        if frozen_weights:
            ops._plane_cache.capture_hold = self._held
        try:
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self._out = model(arg, adj_list)
        finally:
            ops._plane_cache.capture_hold = None

    def __call__(self, x_list=None):
        if x_list is not None:
            xs = [x_list] if self._single else list(x_list)
            if len(xs) != len(self._static_x):
                raise ValueError("GraphedInference: expected %d feature tensors, got %d" % (len(self._static_x), len(xs)))
            for dst, src in zip(self._static_x, xs):
                if dst is src:
                    continue
                if _is_dense(dst) and getattr(dst, "_ctgcn_static", False):
                    raise ValueError("GraphedInference(frozen_weights=True) recorded this static feature tensor in place: capture again for other features")
                if _is_dense(dst):
                    if not _is_dense(src) or src.shape != dst.shape:
                        raise ValueError("GraphedInference: feature shapes are fixed at capture time")
                    dst.copy_(src)
        self.graph.replay()
        return self._out
