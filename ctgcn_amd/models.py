"""CDN / CGCN / CTGCN with the reference's signatures, attribute names (incl. the `duffision` spelling the
checkpoints depend on) and return conventions — reference models.py:8-42, 129-187, 191-253.

CTGCN additionally supports snapshot-parallel execution (one process per GPU, snapshots sharded over
ranks, one all-gather of the per-snapshot hidden states right before the temporal RNN) — see
ctgcn_amd/snapshot_parallel.py.  With no process group the behaviour is the reference's.
"""
import os

import torch
from torch import nn

from .layers import CoreDiffusion, MLP, rnn_reduce_norm
from . import snapshot_parallel as sp_par


class CDN(nn.Module):
    """Stack of CoreDiffusion layers applied to the SAME adjacency list."""

    def __init__(self, input_dim, hidden_dim, output_dim, diffusion_num, bias=True, rnn_type='GRU'):
        super().__init__()
        if diffusion_num < 1:
            raise ValueError("number of layers should be positive!")
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.diffusion_num, self.bias, self.rnn_type = diffusion_num, bias, rnn_type
        dims = [input_dim] + [hidden_dim] * (diffusion_num - 1) + [output_dim]
        self.diffusion_list = nn.ModuleList(
            CoreDiffusion(a, b, bias=bias, rnn_type=rnn_type) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x, adj_list, out=None):
        """out (optional, inference): where the LAST layer writes its [N, output_dim] result (see CoreDiffusion.forward)."""
        last = len(self.diffusion_list) - 1
        for i, layer in enumerate(self.diffusion_list):
            x = layer(x, adj_list, out=out if i == last else None)
        return x


def _check_types(model_type, trans_activate_type):
    assert model_type in ['C', 'S']
    assert trans_activate_type in ['L', 'N']


def _branch_dims(model_type, hidden_dim, output_dim):
    """(mlp output width, CDN input width): 'C' diffuses at hidden width, 'S' at output width."""
    return (hidden_dim, hidden_dim) if model_type == 'C' else (output_dim, output_dim)


class CGCN(nn.Module):
    """Static k-core GCN: one shared MLP + CDN applied to a snapshot or to each snapshot of a list."""

    def __init__(self, input_dim, hidden_dim, output_dim, trans_num, diffusion_num, bias=True, rnn_type='GRU',
                 model_type='C', trans_activate_type='L'):
        super().__init__()
        _check_types(model_type, trans_activate_type)
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.trans_num, self.diffusion_num, self.bias = trans_num, diffusion_num, bias
        self.rnn_type, self.model_type, self.trans_activate_type = rnn_type, model_type, trans_activate_type
        self.method_name = 'CGCN-' + model_type
        mlp_out, cdn_in = _branch_dims(model_type, hidden_dim, output_dim)
        self.mlp = MLP(input_dim, hidden_dim, mlp_out, trans_num, bias=bias, activate_type=trans_activate_type)
        self.duffision = CDN(cdn_in, output_dim, output_dim, diffusion_num, rnn_type=rnn_type)
        self.process_group = None      # set by snapshot_parallel.shard_cgcn()

    def cgcn(self, x, adj):
        trans = self.mlp(x)
        emb = self.duffision(trans, adj)
        return (emb, trans) if self.model_type == 'S' else emb

    def forward(self, x, adj):
        if not isinstance(x, list):
            return self.cgcn(x, adj)
        if self.process_group is not None:
            return sp_par.cgcn_forward_sharded(self, x, adj)
        results = [self.cgcn(x_t, adj_t) for x_t, adj_t in zip(x, adj)]
        if self.model_type == 'C':
            return results
        return [r[0] for r in results], [r[1] for r in results]


class _StackSteps(torch.autograd.Function):
    """torch.stack(hx).transpose(0, 1) made dense in ONE pass: [N, T, d] with out[:, t] = hx[t] (reference models.py:248).  The framework
    route is a cat into [T, N, d], a transposed view and a .contiguous() copy in front of the GRU, and the mirror image in backward; here
    the forward is T strided column writes and the backward hands every snapshot its column of the gradient as a view (the LayerNorm backward
    kernel reads it in place)."""

    @staticmethod
    def forward(ctx, *hx):
        n, d = hx[0].shape
        out = torch.empty(n, len(hx), d, dtype=hx[0].dtype, device=hx[0].device)
        for t, h in enumerate(hx):
            out[:, t].copy_(h)
        return out

    @staticmethod
    def backward(ctx, grad):
        return tuple(grad[:, t] for t in range(grad.shape[1]))


def _stack_steps(hx):
    same = all(h.shape == hx[0].shape and h.dtype == hx[0].dtype and h.device == hx[0].device for h in hx)
    return _StackSteps.apply(*hx) if same and hx[0].dim() == 2 else torch.stack(hx).transpose(0, 1)


class CTGCN(nn.Module):
    """Temporal k-core GCN: per-snapshot MLP + CDN (own weights per snapshot), then a GRU/LSTM over time and
    a LayerNorm.  Returns [T, N, output_dim] ('C') or that plus the per-snapshot MLP outputs ('S')."""

    def __init__(self, input_dim, hidden_dim, output_dim, trans_num, diffusion_num, duration, bias=True, rnn_type='GRU',
                 model_type='C', trans_activate_type='L'):
        super().__init__()
        _check_types(model_type, trans_activate_type)
        assert rnn_type in ['LSTM', 'GRU']
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.rnn_type, self.model_type, self.trans_activate_type = rnn_type, model_type, trans_activate_type
        self.method_name = 'CTGCN-' + model_type
        self.duration, self.trans_num, self.diffusion_num, self.bias = duration, trans_num, diffusion_num, bias
        mlp_out, cdn_in = _branch_dims(model_type, hidden_dim, output_dim)
        self.mlp_list = nn.ModuleList(
            MLP(input_dim, hidden_dim, mlp_out, trans_num, bias=bias, activate_type=trans_activate_type)
            for _ in range(duration))
        self.duffision_list = nn.ModuleList(
            CDN(cdn_in, output_dim, output_dim, diffusion_num, rnn_type=rnn_type) for _ in range(duration))
        rnn_cls = nn.LSTM if rnn_type == 'LSTM' else nn.GRU
        self.rnn = rnn_cls(output_dim, output_dim, num_layers=1, bias=bias, batch_first=True)
        self.norm = nn.LayerNorm(output_dim)
        self.process_group = None      # set by snapshot_parallel.shard_ctgcn()

    def snapshot_branch(self, t, x, adj, out=None):
        """Everything that is independent per snapshot: reference models.py:244-246."""
        trans = self.mlp_list[t](x)
        return self.duffision_list[t](trans, adj, out=out), trans

    def temporal_head(self, hx):
        """hx [N, T, d] -> [T, N, d]: reference models.py:249-250."""
        return rnn_reduce_norm(self.rnn, self.norm, hx, reduce_sum=False).transpose(0, 1)

    def _needs_autograd(self, x_list):
        """True when the forward has to build an autograd graph: grad mode on and a parameter OR an input requires grad
        (frozen weights with x.requires_grad_() — saliency runs, a trainable upstream encoder — must not take the
        write-in-place inference path)."""
        if not torch.is_grad_enabled():
            return False
        return any(p.requires_grad for p in self.parameters()) or any(
            isinstance(x, torch.Tensor) and x.requires_grad for x in x_list)

    def _snapshot_streams(self, seq, n, T, x_list):
        """HIP streams for the snapshot branches of an inference forward, or None.  CTGCN_STREAMS=k asks for k (1 = off); default:
        2 streams for graphs up to 200 000 nodes (launches of tens of microseconds: tails and launch gaps overlap; 4 until the width-128
        layers became one grouped launch per window — since then 2 measure 2 - 5 % faster than 4 or 1 on the four small windows), none above
        (config-5 kernels fill the chip and are HBM-bound: concurrency buys nothing there).
        Only when every dense step of a branch runs in this library's own kernels (GRU width 128, one-hot or split-GEMM-able
        inputs): library GEMMs may use stream-K / split-K kernels whose workgroups wait for each other through flags, and two of
        those running concurrently can occupy every CU with waiting workgroups — measured: the facebook-like CTGCN-S window (its
        1737-wide first Linear went to hipBLASLt) never finished on 4 streams."""
        if seq is None or not seq.is_cuda or T < 2:
            return None
        env = os.environ.get("CTGCN_STREAMS")
        k = int(env) if env else (2 if n <= 200_000 else 1)
        k = min(k, T)
        if k <= 1 or not self._branches_use_own_kernels(x_list):
            return None
        cache = getattr(self, "_stream_cache", None)
        if cache is None or cache[0] != (seq.device, k):
            cache = self._stream_cache = ((seq.device, k), [torch.cuda.Stream(device=seq.device) for _ in range(k)])
        return cache[1]

    def _training_streams(self, x_list, adj_list, T):
        """HIP streams for the snapshot branches of a TRAINING forward (and, through autograd, of its backward), or None.
        CTGCN_TRAIN_STREAMS=k (1 = off; default 3: measured Enron-like 115 -> 103 ms per training step, math-like 38 -> 34, AS-like
        17 -> 14, profiles/r05_train_small.txt; the gradients are the same bits).  Only where every dense step of a branch and of its backward runs in this library's
        kernels — one-hot features, GRU width 128, fp16x2 arithmetic (see _snapshot_streams on concurrent library GEMMs) — and the graph is
        small (<= 200 000 nodes)."""
        from . import ops
        from .layers import _is_identity
        k = min(int(os.environ.get("CTGCN_TRAIN_STREAMS", "3")), T)
        if k <= 1 or self.rnn_type != 'GRU' or self.output_dim != 128 or not ops.split_mfma_enabled() or not ops.linear_split_enabled() \
                or not ops.wide_weight_grad_enabled() or torch.cuda.is_current_stream_capturing():
            return None
        p0 = next(self.parameters())
        if not p0.is_cuda or not all(torch.is_tensor(x) and x.is_sparse and x.is_cuda and _is_identity(x) for x in x_list):
            return None
        n = adj_list[0].n if hasattr(adj_list[0], "n") else adj_list[0][0].shape[0]
        if n > 200_000 or any(m.layer_num != 1 for m in self.mlp_list):
            return None
        cache = getattr(self, "_train_stream_cache", None)
        if cache is None or cache[0] != (p0.device, k):
            cache = self._train_stream_cache = ((p0.device, k), [torch.cuda.Stream(device=p0.device) for _ in range(k)])
        return cache[1]

    def _branches_use_own_kernels(self, x_list):
        """The verdict is cached on what it depends on — the inputs' KIND (identity or not, checked by layers._is_identity, whose own cache
        keeps the tensor alive), shape, strides and alignment — never on id(x): a recycled id with another tensor must not reuse it."""
        from . import ops
        from .layers import _is_identity

        def kind(x):
            if not torch.is_tensor(x):
                return None
            if x.is_sparse:
                return ("sparse", _is_identity(x), tuple(x.shape))
            return ("dense", tuple(x.shape), tuple(x.stride()), x.data_ptr() % 16, x.dtype, x.device)
        key = (ops.forward_split_mode(), ops.linear_split_enabled()) + tuple(kind(x) for x in x_list)
        cache = getattr(self, "_own_kernel_cache", None)
        if cache is None or cache[0] != key:
            cache = self._own_kernel_cache = (key, self._branches_use_own_kernels_uncached(x_list))
        return cache[1]

    def _branches_use_own_kernels_uncached(self, x_list):
        """Decided by the dispatch predicates the branch itself uses (ops.linear_split_ok on the real first operand, on a probe row of
        the right width for the layers behind it — outputs of this library's kernels are fresh contiguous tensors)."""
        from . import ops
        from .layers import _is_identity
        if self.rnn_type != 'GRU' or self.output_dim != 128 or ops.forward_split_mode() != 2 or not ops.linear_split_enabled():
            return False
        for t, x in enumerate(x_list):
            if not torch.is_tensor(x) or not x.is_cuda:
                return False
            mlp = self.mlp_list[t]
            layers = [mlp.linear] if mlp.layer_num == 1 else list(mlp.linears)
            if x.is_sparse:
                if not _is_identity(x) or layers[0].weight.dtype != torch.float32:
                    return False                                # generic sparse features: torch.sparse.mm
                layers = layers[1:]                             # Linear(I) = W^T + b: transpose kernel
                h = None
                width = mlp.output_dim if mlp.layer_num == 1 else mlp.hidden_dim
            else:
                h, width = x, x.shape[1]
            for lin in layers:                                  # dense Linear layers must be taken by ctgcn_linear_f32 (MLP._apply_linear)
                probe = h if h is not None else torch.empty(1, width, dtype=torch.float32, device=x.device)
                if probe.dim() != 2 or not ops.linear_split_ok(probe, lin.weight):
                    return False
                h, width = None, lin.weight.shape[0]
            for cd in self.duffision_list[t].diffusion_list:    # CoreDiffusion GRU input width: 128 (resident-weight kernels) or split-GEMM-able
                if cd.input_dim % 4 or cd.input_dim < 32 or cd.input_dim > 512 or not cd.rnn.weight_ih_l0.is_contiguous():
                    return False                                # ops.aggregate_split_ok would send it to the fp32 H + library path
        return True

    def _group_start(self, adj_list, T):
        """Index of the first CoreDiffusion layer from which every later layer of every snapshot is 128 -> 128 (the shape of the grouped
        launches, ops.core_diffusion_split_group), or None: 'C' configs -> their second layer, 'S' -> their only one."""
        from . import ops
        if not ops.group_launch_enabled() or T < 2 or self.rnn_type != 'GRU' or self.output_dim != 128 or not hasattr(adj_list[0], "n"):
            return None
        if adj_list[0].n > ops._GROUP_MAX_NODES:
            return None
        L = self.diffusion_num
        g0 = L
        for l in reversed(range(L)):
            if all(cdn.diffusion_list[l].input_dim == 128 and cdn.diffusion_list[l].output_dim == 128 for cdn in self.duffision_list):
                g0 = l
            else:
                break
        return g0 if g0 < L else None

    _HEAD_GROUP_MAX = 120_000      # nodes x snapshots up to which the first layer runs as grouped launches

    def _grouped_head(self, x_list, adj_list, seq):
        """(first-layer outputs, MLP outputs) of every snapshot with the first CoreDiffusion layer (d_in != 128) of the window in one
        launch per kernel (ops.core_diffusion_wide_group), or None when the window does not fit that path.  The MLPs in front: one transpose
        launch when every snapshot's is a single Linear on one-hot features without activation, else per snapshot.
        Only for the smallest windows (nodes x snapshots <= _HEAD_GROUP_MAX; CTGCN_GROUP_HEAD=1 / 0 forces / forbids it): measured
        (profiles/r05_group_head.txt) AS-like 1.87 -> 1.32 ms per window, math-like 4.70 -> 5.01, Enron-like 14.5 -> 15.4 — at 0.2 - 0.5 ms
        per kernel the per-snapshot launches on two lanes overlap one snapshot's HBM-bound aggregation with another's GEMM and recurrence,
        which one launch per kernel cannot (neither can two half-window groups on two lanes: every launch already fills all CUs)."""
        from . import ops
        from .layers import as_core_adj, _is_identity
        T = len(x_list)
        dev = seq.device
        mods = [self.duffision_list[t].diffusion_list[0] for t in range(T)]
        if any(m.input_dim == 128 or m.output_dim != 128 for m in mods) or not ops.group_launch_enabled():
            return None
        adjs = [as_core_adj(adj_list[t], dev) for t in range(T)]
        env = os.environ.get("CTGCN_GROUP_HEAD")
        if env == "0" or (env != "1" and adjs[0].n * T > self._HEAD_GROUP_MAX):
            return None
        mlps = [self.mlp_list[t] for t in range(T)]
        one_hot = all(m.layer_num == 1 and m.activate_type == 'L' and torch.is_tensor(x) and x.is_sparse and x.is_cuda and _is_identity(x)
                      for m, x in zip(mlps, x_list))
        ws, bs = [m.linear.weight for m in mlps] if one_hot else None, [m.linear.bias for m in mlps] if one_hot else None
        if one_hot and ops.linear_of_identity_group_ok(ws, bs):
            trans = ops.linear_of_identity_group(ws, bs)
        else:
            trans = [mlps[t](x_list[t]) for t in range(T)]
        rnns, norms = [m.rnn for m in mods], [m.norm for m in mods]
        if ops.core_diffusion_wide_group_ok(trans, adjs, rnns, norms):
            outs = [torch.empty(trans[t].shape[0], self.output_dim, dtype=trans[t].dtype, device=dev) for t in range(T)]
            ops.core_diffusion_wide_group(trans, adjs, rnns, norms, outs)
        else:
            outs = [mods[t](trans[t], adjs[t]) for t in range(T)]
        return outs, trans

    def _grouped_layers(self, g0, hs, adj_list, seq):
        """CoreDiffusion layers g0 .. of every snapshot, one aggregation launch + one GRU launch per layer for the whole window (reference
        models.py:243-247 loops over the snapshots); the last layer writes column t of the temporal GRU's input."""
        from . import ops
        from .layers import as_core_adj
        T, L = len(hs), self.diffusion_num
        adjs = [as_core_adj(adj_list[t], hs[t].device) for t in range(T)]
        for l in range(g0, L):
            mods = [self.duffision_list[t].diffusion_list[l] for t in range(T)]
            last = l == L - 1
            outs = [seq[:, t] if last else torch.empty(hs[t].shape[0], self.output_dim, dtype=hs[t].dtype, device=hs[t].device) for t in range(T)]
            rnns, norms = [m.rnn for m in mods], [m.norm for m in mods]
            if ops.core_diffusion_group_ok(hs, adjs, rnns, norms):
                ops.core_diffusion_split_group(hs, adjs, rnns, norms, outs)
            else:                                        # e.g. a hub row in one snapshot: that window takes the per-snapshot launches
                for t in range(T):
                    r = mods[t](hs[t], adjs[t], out=outs[t] if last else None)
                    if last and r.data_ptr() != outs[t].data_ptr():
                        outs[t].copy_(r)
                    elif not last:
                        outs[t] = r
            hs = outs
        return hs

    def forward(self, x_list, adj_list):
        if self.process_group is not None:
            return sp_par.ctgcn_forward_sharded(self, x_list, adj_list)
        hx, trans = [], []
        T = len(x_list)
        seq = None
        if T > 0 and not self._needs_autograd(x_list):
            # inference: every snapshot's embeddings are written straight into column t of the temporal GRU's [N, T, d]
            # input (models.py:248 stack + transpose without the two copies)
            n = adj_list[0].n if hasattr(adj_list[0], "n") else adj_list[0][0].shape[0]
            p0 = next(self.parameters())
            seq = torch.empty(n, T, self.output_dim, dtype=p0.dtype, device=p0.device)
        lanes = self._snapshot_streams(seq, n if seq is not None else 0, T, x_list)
        g0 = self._group_start(adj_list, T) if (seq is not None and seq.is_cuda) else None
        grouped_head = self._grouped_head(x_list, adj_list, seq) if g0 == 1 else None
        if grouped_head is not None:
            # small 'C' window: Linear(I), then the 500-wide first layer of ALL snapshots in one launch per kernel, then the width-128 layers
            hx, trans = grouped_head
            self._grouped_layers(g0, hx, adj_list, seq)
        elif g0 is not None:
            # small window: the MLP and the layers before g0 per snapshot (on the lanes), then the width-128 layers of ALL snapshots per launch
            def head(t):
                tr = self.mlp_list[t](x_list[t])
                h = tr
                for l in range(g0):
                    h = self.duffision_list[t].diffusion_list[l](h, adj_list[t])
                return h, tr
            if lanes:
                main = torch.cuda.current_stream(seq.device)
                for s_ in lanes:
                    s_.wait_stream(main)
                for t in range(T):
                    with torch.cuda.stream(lanes[t % len(lanes)]):
                        h, tr = head(t)
                        h.record_stream(main)
                        tr.record_stream(main)
                    hx.append(h)
                    trans.append(tr)
                for s_ in lanes:
                    main.wait_stream(s_)
            else:
                for t in range(T):
                    h, tr = head(t)
                    hx.append(h)
                    trans.append(tr)
            self._grouped_layers(g0, hx, adj_list, seq)
        elif lanes:
            # inference on a small graph: the snapshot branches are independent until the temporal GRU (models.py:243-247) and their
            # kernels are too short to fill the chip one after the other — run them on a few HIP streams, join before the head
            main = torch.cuda.current_stream(seq.device)
            for s_ in lanes:
                s_.wait_stream(main)
            for t in range(T):
                with torch.cuda.stream(lanes[t % len(lanes)]):
                    h, tr = self.snapshot_branch(t, x_list[t], adj_list[t], out=seq[:, t])
                    if h.data_ptr() != seq[:, t].data_ptr():
                        seq[:, t].copy_(h)
                    tr.record_stream(main)                 # returned to the caller ('S'), who uses it on the main stream
                hx.append(h)
                trans.append(tr)
            for s_ in lanes:
                main.wait_stream(s_)
        elif seq is None and self._training_streams(x_list, adj_list, T):
            # training on a small window: the snapshot branches on a few HIP streams, like the inference lanes — autograd runs a node's
            # backward on the stream of its forward, so the backward kernels of different snapshots overlap as well
            tl = self._training_streams(x_list, adj_list, T)
            main = torch.cuda.current_stream(tl[0].device)
            for s_ in tl:
                s_.wait_stream(main)
            for t in range(T):
                with torch.cuda.stream(tl[t % len(tl)]):
                    h, tr = self.snapshot_branch(t, x_list[t], adj_list[t])
                    h.record_stream(main)
                    tr.record_stream(main)
                hx.append(h)
                trans.append(tr)
            for s_ in tl:
                main.wait_stream(s_)
        else:
            for t in range(T):
                h, tr = self.snapshot_branch(t, x_list[t], adj_list[t], out=None if seq is None else seq[:, t])
                if seq is not None and h.data_ptr() != seq[:, t].data_ptr():
                    seq[:, t].copy_(h)           # a layer that could not write in place (other rnn type/width) returned its own tensor
                hx.append(h)
                trans.append(tr)
        out = self.temporal_head(seq if seq is not None else _stack_steps(hx))
        return out if self.model_type == 'C' else (out, trans)
