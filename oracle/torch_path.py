"""torch_path.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's *CPU torch.sparse path* restated with stock PyTorch, driven purely by a
state_dict (no module classes shared with ctgcn_amd).  Used as (i) the model-level parity
oracle on seeded inputs and (ii) bench.py's cpu_baseline ("kind": "port").

Parity status: PINNED — tests/test_oracle_golden.py replays tests/golden/models_uci.npz and
weighted_small.npz (outputs of the reference itself) through these functions.

Citations are to the reference tree.
"""
import numpy as np
import torch
import torch.nn.functional as F


def coo_like_reference(csr):
    """utils.py:89-95: scipy matrix -> torch sparse COO, int64 indices, fp32 values, NOT coalesced."""
    coo = csr.tocoo()
    idx = torch.from_numpy(np.vstack((coo.row, coo.col))).long()
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data).float(), torch.Size(coo.shape))


_tap = None      # tests: a list that receives, per aggregate_loop call, {"pre": [K detached pre-ReLU tensors]} (cdn_rows adds "rows")


def aggregate_loop(adj_list, x):
    """layers.py:41-48: the K-step cumulative torch.sparse.mm loop + ReLU. Returns list of K [N,d]."""
    acc, out = None, []
    for a in adj_list:
        y = torch.sparse.mm(a, x)
        acc = y if acc is None else acc + y
        out.append(acc)
    if _tap is not None:
        _tap.append({"pre": [v.detach() for v in out]})
    return [F.relu(v) for v in out]


def _rnn(sd, prefix, rnn_type, seq):
    """nn.GRU / nn.LSTM(num_layers=1, batch_first=True) evaluated from raw weights (layers.py:27-30).  On a GPU the rows go through the
    module in chunks: MIOpen rejects problems whose gate buffers pass 2^31 elements (1 M nodes x 8 cores x 3 x 128 already does); rows are
    independent sequences, so this is the same result."""
    w_ih, w_hh = sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"]
    b_ih, b_hh = sd.get(prefix + "bias_ih_l0"), sd.get(prefix + "bias_hh_l0")
    hid = w_hh.shape[1]
    mod = (torch.nn.LSTM if rnn_type == "LSTM" else torch.nn.GRU)(w_ih.shape[1], hid, 1, bias=b_ih is not None,
                                                                 batch_first=True).to(device=w_ih.device, dtype=w_ih.dtype)   # float64 runs = exact truth
    with torch.no_grad():
        mod.weight_ih_l0.copy_(w_ih)
        mod.weight_hh_l0.copy_(w_hh)
        if b_ih is not None:
            mod.bias_ih_l0.copy_(b_ih)
            mod.bias_hh_l0.copy_(b_hh)
    for p in mod.parameters():
        p.requires_grad_(False)
    chunk = (1 << 29) // max(1, seq.shape[1] * 4 * hid)
    if not seq.is_cuda or seq.shape[0] <= chunk:
        return mod(seq)[0]
    return torch.cat([mod(seq[lo:lo + chunk])[0] for lo in range(0, seq.shape[0], chunk)], 0)


def _rnn_grad(sd, prefix, rnn_type, seq):
    """Same recurrence written out with torch ops so that gradients flow to the tensors in `sd`
    (PyTorch's documented GRU/LSTM cell equations, gate order r,z,n / i,f,g,o)."""
    w_ih, w_hh = sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"]
    b_ih, b_hh = sd.get(prefix + "bias_ih_l0"), sd.get(prefix + "bias_hh_l0")
    hid = w_hh.shape[1]
    h = seq.new_zeros(seq.shape[0], hid)
    c = seq.new_zeros(seq.shape[0], hid)
    outs = []
    for t in range(seq.shape[1]):
        gi = F.linear(seq[:, t], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        if rnn_type == "LSTM":
            i, f, g, o = (gi + gh).chunk(4, 1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
        else:
            ir, iz, inn = gi.chunk(3, 1)
            hr, hz, hn = gh.chunk(3, 1)
            r, z = torch.sigmoid(ir + hr), torch.sigmoid(iz + hz)
            nn_ = torch.tanh(inn + r * hn)
            h = (1 - z) * nn_ + z * h
        outs.append(h)
    return torch.stack(outs, 1)


def ctgcn_with_grad(sd, x_list, adj_list, rnn_type="GRU", model_type="C", activate="L"):
    """ctgcn() with every recurrence unrolled in torch ops: differentiable w.r.t. the tensors of `sd`."""
    global _rnn
    saved = _rnn
    _rnn = _rnn_grad
    try:
        return ctgcn(sd, x_list, adj_list, rnn_type, model_type, activate)
    finally:
        _rnn = saved


def cgcn_with_grad(sd, x, adj, rnn_type="GRU", model_type="C", activate="L"):
    """cgcn() with every recurrence unrolled in torch ops: differentiable w.r.t. the tensors of `sd`."""
    global _rnn
    saved = _rnn
    _rnn = _rnn_grad
    try:
        return cgcn(sd, x, adj, rnn_type, model_type, activate)
    finally:
        _rnn = saved


def core_diffusion(sd, prefix, x, adj_list, rnn_type="GRU"):
    """layers.py:38-63."""
    hs = aggregate_loop(adj_list, x)
    seq = torch.stack(hs, 0).transpose(0, 1)
    out = _rnn(sd, prefix + "rnn.", rnn_type, seq).sum(1)
    w, b = sd[prefix + "norm.weight"], sd[prefix + "norm.bias"]
    return F.layer_norm(out, (w.shape[0],), w, b)


def cdn(sd, prefix, x, adj_list, rnn_type="GRU"):
    """models.py:39-42."""
    n = 0
    while (prefix + "diffusion_list.%d.norm.weight" % n) in sd:
        n += 1
    for l in range(n):
        x = core_diffusion(sd, prefix + "diffusion_list.%d." % l, x, adj_list, rnn_type)
    return x


def mlp(sd, prefix, x, activate):
    """layers.py:95-106 (x may be a sparse COO identity, helper.py:169-171)."""
    def lin(p, h):
        w, b = sd[p + "weight"], sd.get(p + "bias")
        h = torch.sparse.mm(h, w.t()) if h.is_sparse else h @ w.t()
        h = h + b if b is not None else h
        return F.selu(h) if activate == "N" else h
    if (prefix + "linear.weight") in sd:
        return lin(prefix + "linear.", x)
    i = 0
    while (prefix + "linears.%d.weight" % i) in sd:
        x = lin(prefix + "linears.%d." % i, x)
        i += 1
    return x


def cgcn(sd, x, adj, rnn_type="GRU", model_type="C", activate="L"):
    """models.py:165-187."""
    def one(xx, aa):
        tr = mlp(sd, "mlp.", xx, activate)
        return cdn(sd, "duffision.", tr, aa, rnn_type), tr
    if isinstance(x, list):
        res = [one(xx, aa) for xx, aa in zip(x, adj)]
        emb = [r[0] for r in res]
        return emb if model_type == "C" else (emb, [r[1] for r in res])
    e, tr = one(x, adj)
    return e if model_type == "C" else (e, tr)


def ctgcn(sd, x_list, adj_list, rnn_type="GRU", model_type="C", activate="L"):
    """models.py:240-253."""
    hx, trans = [], []
    for t in range(len(x_list)):
        tr = mlp(sd, "mlp_list.%d." % t, x_list[t], activate)
        trans.append(tr)
        hx.append(cdn(sd, "duffision_list.%d." % t, tr, adj_list[t], rnn_type))
    seq = torch.stack(hx).transpose(0, 1)
    out = _rnn(sd, "rnn.", rnn_type, seq)
    w, b = sd["norm.weight"], sd["norm.bias"]
    out = F.layer_norm(out, (w.shape[0],), w, b).transpose(0, 1)
    return out if model_type == "C" else (out, trans)


# ----------------------------------------------------------------------------------- rows of the outputs
# The reference path evaluated for a SUBSET of the nodes — for parity checks at sizes where the whole float64 forward + autograd
# would take minutes (1 M nodes).  Nothing new is computed: rows R of torch.sparse.mm(A, x) are torch.sparse.mm(A[R, :], x), and
# the GRU / LayerNorm behind it (layers.py:59-62) act on every row by itself, so rows R of a CoreDiffusion layer's output are
# core_diffusion() called with the row-sliced matrices.  A stack of layers (models.py:39-42) needs the layer before it on the
# columns those sliced matrices touch — worked out back to front, scattered into an otherwise-zero [N, d] operand front to back.
# Autograd runs through all of it, so the gradient of a loss that only reads rows R (the reference's batch loss reads the batch's
# nodes, their walk partners and the negatives: embedding.py:346-352, metrics.py:38-60) is the same as through the full forward.  Pinned against ctgcn() / ctgcn_with_grad() on whole
# small graphs by tests/test_oracle_golden.py::test_row_subset_path_equals_the_full_path.
# ReLU kinks: a pre-activation within rounding of zero has a derivative that depends on the arithmetic (0 or 1); `_tap` hands the tests the
# float64 pre-activations so that they can keep such entries out of the loss (ambiguous_rows below) instead of widening tolerances.
def ambiguous_rows(pre_list, rel=1e-5):
    """bool [rows]: some pre-ReLU value of the row (any core, any feature) lies within rel x the row's largest |value| of zero"""
    stack = torch.stack(pre_list, 1).abs()                              # [rows, K, d]
    big = stack.reshape(stack.shape[0], -1).max(1).values                # relative to the row's own scale: the sum's rounding error is
    return (stack.reshape(stack.shape[0], -1).min(1).values <= rel * big)  # a few ulps of its largest partial sums, whatever their magnitude


def _rows_of(mats, rows, dtype):
    """[A[rows, :] for A in mats] as the reference's COO tensors (utils.py:89-95); mats: scipy matrices"""
    import scipy.sparse as sp
    return [coo_like_reference(sp.csr_matrix(m)[rows]).to(dtype) for m in mats]


def _columns_of(mats, rows):
    """sorted union of `rows` (the + I of the first matrix, helper.py:71-72) and every column the rows touch in any matrix"""
    import scipy.sparse as sp
    cols = [np.asarray(rows, dtype=np.int64)]
    for m in mats:
        cols.append(sp.csr_matrix(m)[rows].indices.astype(np.int64))
    return np.unique(np.concatenate(cols))


def cdn_rows(sd, prefix, x, mats, rows, rnn_type="GRU"):
    """rows `rows` of cdn(sd, prefix, x, adj_list) (models.py:39-42); x: the full [N, d] input, mats: scipy k-core list"""
    n_layers = 0
    while (prefix + "diffusion_list.%d.norm.weight" % n_layers) in sd:
        n_layers += 1
    need = [np.asarray(rows, dtype=np.int64)]
    for _ in range(n_layers - 1):
        need.insert(0, _columns_of(mats, need[0]))
    for l in range(n_layers):
        y = core_diffusion(sd, prefix + "diffusion_list.%d." % l, x, _rows_of(mats, need[l], x.dtype), rnn_type)
        if _tap is not None:
            _tap[-1]["rows"] = need[l]
        if l + 1 < n_layers:
            x = torch.zeros(mats[0].shape[0], y.shape[1], dtype=y.dtype).index_put((torch.from_numpy(need[l]),), y)
    return y


def ctgcn_rows(sd, x_list, mats_list, rows, rnn_type="GRU", model_type="C", activate="L", with_grad=False):
    """ctgcn(...)[:, rows] (models.py:240-253) -> [T, len(rows), d]; mats_list: per snapshot the scipy k-core list"""
    global _rnn
    saved = _rnn
    if with_grad:
        _rnn = _rnn_grad
    try:
        hx = []
        for t in range(len(x_list)):
            x, pre = x_list[t], "duffision_list.%d." % t
            if x.is_sparse:
                tr = mlp(sd, "mlp_list.%d." % t, x, activate)
            else:
                # dense features (CTGCN-S: width 1 737 through three Linear layers): the MLP acts row by row (layers.py:95-106), so only the
                # rows the CoreDiffusion layers will read are transformed — `rows` and every column their slices touch, layer by layer
                n_layers = 0
                while (pre + "diffusion_list.%d.norm.weight" % n_layers) in sd:
                    n_layers += 1
                need = np.asarray(rows, dtype=np.int64)
                for _ in range(n_layers):
                    need = _columns_of(mats_list[t], need)
                idx = torch.from_numpy(need)
                part = mlp(sd, "mlp_list.%d." % t, x[idx], activate)
                tr = torch.zeros(x.shape[0], part.shape[1], dtype=part.dtype).index_put((idx,), part)
            hx.append(cdn_rows(sd, pre, tr, mats_list[t], rows, rnn_type))
        seq = torch.stack(hx).transpose(0, 1)
        out = _rnn(sd, "rnn.", rnn_type, seq)
        w, b = sd["norm.weight"], sd["norm.bias"]
        return F.layer_norm(out, (w.shape[0],), w, b).transpose(0, 1)
    finally:
        _rnn = saved
