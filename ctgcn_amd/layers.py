"""CoreDiffusion and MLP layers with the reference's constructor/forward signatures and state_dict keys
(reference layers.py:9-63 and :67-106), running the sparse aggregation on the HIP kernel.

The aggregation (reference layers.py:41-48 + the stack/transpose of :58) runs in ctgcn_core_aggregate_f32; the
core-axis GRU/LSTM + sum over cores + LayerNorm (layers.py:59-62) in the fused recurrence kernels when hidden = 128
(ops.gru_sequence / ops.lstm_sequence), otherwise in the PyTorch-ROCm modules, chunked over rows.
"""
import torch
from torch import nn
from torch.nn import functional as F

from . import ops
from .core_adj import CoreAdj

_RNN = {"GRU": nn.GRU, "LSTM": nn.LSTM}
_RNN_MAX_ELEMS = 1 << 29     # MIOpen's RNN indexes its gate workspace with 32-bit ints: keep batch*seq*4*hidden below 2^31


def rnn_reduce_norm(rnn, norm, seq, reduce_sum, out=None):
    """norm(rnn(seq).sum(1)) or norm(rnn(seq)).  GRU(hidden=128): fused HIP kernels, forward and backward.
    LSTM(hidden=128): fused HIP recurrence for inference.  Everything else (other widths, LSTM training) goes through
    the PyTorch-ROCm modules.  out (inference only, reduce_sum): a [rows, hidden] strided view that receives the result."""
    if out is not None and torch.is_grad_enabled() and (seq.requires_grad or any(p.requires_grad for p in rnn.parameters())):
        out = None                                   # training: autograd needs its own tensors
    if ops.gru_fused_ok(rnn, seq):
        if out is not None and reduce_sum:
            return ops.gru_sequence(rnn, seq, norm, reduce_sum, out=out)
        res = ops.gru_sequence(rnn, seq, norm, reduce_sum)
    elif ops.lstm_fused_ok(rnn, seq):
        res = ops.lstm_sequence(rnn, seq, norm, reduce_sum)
    else:
        res = norm(rnn_over_rows(rnn, seq, reduce_sum))
    if out is not None:
        out.copy_(res)
        return out
    return res


def rnn_over_rows(rnn, seq, reduce_sum):
    """rnn(seq)[0] for seq [rows, steps, feat], evaluated in row chunks (rows are independent sequences, so this is
    exact).  MIOpen rejects (miopenStatusBadParm) problems whose gate buffers exceed 2^31 elements — 1M nodes x 8
    cores x 3*128 already does.  With reduce_sum the per-row sum over steps is taken chunk by chunk, so the
    [rows, steps, hidden] output of the full problem is never materialised."""
    rows, steps, _ = seq.shape
    gates = 4 if isinstance(rnn, nn.LSTM) else 3
    chunk = max(1, _RNN_MAX_ELEMS // max(1, steps * gates * rnn.hidden_size))
    if rows <= chunk:
        out = rnn(seq)[0]
        return out.sum(dim=1) if reduce_sum else out
    parts = []
    for lo in range(0, rows, chunk):
        out = rnn(seq[lo:lo + chunk])[0]
        parts.append(out.sum(dim=1) if reduce_sum else out)
    return torch.cat(parts, 0)
_adj_cache = {}


def as_core_adj(adj_list, device):
    """Accept what the reference's callers pass: a CoreAdj (our loader) or a python list of torch sparse
    matrices (reference loader, helper.py:51-82).  Lists are fused once and cached by identity."""
    if isinstance(adj_list, CoreAdj):
        return adj_list if adj_list.device == device else adj_list.to(device)    # .to() keeps the moved copy on the source object
    key = tuple((id(a), a._values().data_ptr() if a.is_sparse else a.data_ptr()) for a in adj_list) + (str(device),)
    hit = _adj_cache.get(key)
    if hit is None:
        if len(_adj_cache) > 64:
            _adj_cache.clear()
        # the entry keeps the source tensors alive, so their ids cannot be recycled while the entry exists
        hit = _adj_cache[key] = (CoreAdj.from_matrices(list(adj_list), device=device), list(adj_list))
    return hit[0]


_identity_cache = {}


def _is_identity(sp_tensor):
    """True iff the sparse COO tensor is exactly the N x N identity (what get_feature_list builds for one-hot
    node features).  Checked once per tensor (one small reduction + host read), then cached by identity; the
    cache entry holds the tensor, so its id cannot be recycled while the entry exists."""
    key = (id(sp_tensor), sp_tensor._values().data_ptr())
    entry = _identity_cache.get(key)
    if entry is None:
        n, m = sp_tensor.shape
        idx, val = sp_tensor._indices(), sp_tensor._values()
        flag = bool(n == m and val.numel() == n and idx.shape[0] == 2
                    and bool(((idx[0] == idx[1]) & (val == 1)).all())
                    and bool((idx[0].sort().values == torch.arange(n, device=idx.device)).all()))
        if len(_identity_cache) > 256:
            _identity_cache.clear()
        entry = _identity_cache[key] = (flag, sp_tensor)
    return entry[0]


class CoreDiffusion(nn.Module):
    """K-core diffusion layer: H_j = relu(sum_{i<=j} A_i x), RNN over j, sum over j, LayerNorm."""

    def __init__(self, input_dim, output_dim, core_num=1, bias=True, rnn_type='GRU'):
        super().__init__()
        assert rnn_type in _RNN
        self.input_dim, self.output_dim = input_dim, output_dim
        self.core_num, self.bias, self.rnn_type = core_num, bias, rnn_type
        # `linear` is never used in forward (nor in the reference, layers.py:24) but is part of the checkpoint schema
        self.linear = nn.Linear(input_dim, output_dim)
        self.rnn = _RNN[rnn_type](input_size=input_dim, hidden_size=output_dim, num_layers=1, bias=bias, batch_first=True)
        self.norm = nn.LayerNorm(output_dim)

    def aggregate(self, x, adj_list):
        """[N, K, input_dim] = all K rectified cumulative aggregations, one kernel launch."""
        return ops.core_aggregate(x, as_core_adj(adj_list, x.device), relu=True)

    def forward(self, x, adj_list, out=None):
        """out (optional, inference): a [N, output_dim] view (unit column stride) that receives the result."""
        if torch.is_tensor(x) and not x.is_sparse and x.is_cuda:
            adj = as_core_adj(adj_list, x.device)
            if ops.aggregate_split_ok(self.rnn, x, adj):
                # inference, GRU input projection on the split GEMM (d_in != 128): the aggregation writes the GEMM's fp16 operand
                # planes, the fp32 [N, K, input_dim] tensor is never materialised
                return ops.core_diffusion_split(x, adj, self.rnn, self.norm, out=out)
            if torch.is_grad_enabled() and ops.core_diffusion_fused_ok(self.rnn, self.norm, x, adj):
                # training, d_in = hidden = 128: the inference forward under autograd (planes + row plan kept for the backward kernels)
                return ops.core_diffusion_fused(x, adj, self.rnn, self.norm)
        seq = self.aggregate(x, adj_list)            # [batch = N, seq = K, feat]
        return rnn_reduce_norm(self.rnn, self.norm, seq, reduce_sum=True, out=out)


class MLP(nn.Module):
    """Stack of Linear layers, SELU after each when activate_type == 'N' (reference layers.py:67-106).
    Key names: `linear.*` for one layer, `linears.{i}.*` otherwise."""

    def __init__(self, input_dim, hidden_dim, output_dim, layer_num, bias=True, activate_type='N'):
        super().__init__()
        assert activate_type in ['L', 'N']
        assert layer_num > 0
        self.input_dim, self.hidden_dim, self.output_dim = input_dim, hidden_dim, output_dim
        self.layer_num, self.bias, self.activate_type = layer_num, bias, activate_type
        if layer_num == 1:
            self.linear = nn.Linear(input_dim, output_dim, bias=bias)
        else:
            widths = [input_dim] + [hidden_dim] * (layer_num - 1) + [output_dim]
            self.linears = nn.ModuleList(nn.Linear(a, b, bias=bias) for a, b in zip(widths[:-1], widths[1:]))

    @staticmethod
    def _apply_linear(layer, h, selu=False, static_x=False):
        """layer(h), followed by F.selu when asked (fused into the split GEMM's epilogue on that path)"""
        out, fused = MLP._linear(layer, h, selu, static_x)
        return F.selu(out) if (selu and not fused) else out

    @staticmethod
    def _linear(layer, h, selu, static_x=False):
        """(layer(h), whether the activation was already applied)"""
        if h.is_sparse:          # one-hot / sparse features (helper.py:161-172)
            if _is_identity(h):  # Linear(I) = W^T + b: one pass over W instead of an N x N SpMM
                if h.is_cuda and layer.weight.dtype == torch.float32:
                    return ops.linear_of_identity(layer.weight, layer.bias), False      # HIP transpose, forward and backward
                return (layer.weight.t() + layer.bias if layer.bias is not None else layer.weight.t().contiguous()), False
            out = torch.sparse.mm(h, layer.weight.t())
            return (out if layer.bias is None else out + layer.bias), False
        needs_grad = torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in layer.parameters()))
        if not needs_grad and h.dim() == 2 and ops.linear_split_ok(h, layer.weight):
            # fp32-accurate split GEMM on the 16-bit matrix cores (inference), SELU in its epilogue
            return ops.linear_split(h, layer.weight, layer.bias, selu=selu, static_x=static_x), bool(selu)
        if needs_grad and h.dim() == 2 and ops.linear_train_enabled() and ops.linear_split_ok(h, layer.weight) and ops.plane_cache_enabled() \
                and h.shape[0] >= 4096:
            # training: forward and dx on the same GEMM (ops._LinearSplit); small inputs stay with the library (launch-bound either way)
            return ops.linear_split_train(h, layer.weight, layer.bias, static_x=static_x), False
        return layer(h), False

    def forward(self, x):
        stack = [self.linear] if self.layer_num == 1 else list(self.linears)
        selu = self.activate_type == 'N'
        if len(stack) > 1 and not (torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or
                                                              (torch.is_tensor(x) and x.requires_grad))):
            # inference through dense layers: a hidden activation leaves its GEMM as the next GEMM's operand planes (ops.linear_chain_ok /
            # ctgcn_linear_packed_chain_f32) — no fp32 rows written, read back and split between two layers
            i = 0
            while i + 1 < len(stack) and ops.linear_chain_ok(x, stack[i].weight, stack[i + 1].weight):
                x = ops.linear_split(x, stack[i].weight, stack[i].bias, selu=selu, static_x=(i == 0 and ops.is_static(x)), planes_out=True)
                i += 1
            if i:
                x = ops.linear_split(x, stack[i].weight, stack[i].bias, selu=selu)       # the chain's last GEMM: fp32 rows
                stack = stack[i + 1:]
        for i, layer in enumerate(stack):
            # the first layer's operand is the module's input: when the caller has declared it static (ops.mark_static — node features
            # built once and passed to every forward, train.py:72-76, embedding.py:318; ctgcn_amd.helper.DataLoader does) its operand
            # planes are kept between calls (ops._PlaneCache, validated by the version counter)
            x = self._apply_linear(layer, x, selu=self.activate_type == 'N', static_x=(i == 0 and ops.is_static(x)))
        return x
