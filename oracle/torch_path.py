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


def aggregate_loop(adj_list, x):
    """layers.py:41-48: the K-step cumulative torch.sparse.mm loop + ReLU. Returns list of K [N,d]."""
    acc, out = None, []
    for a in adj_list:
        y = torch.sparse.mm(a, x)
        acc = y if acc is None else acc + y
        out.append(acc)
    return [F.relu(v) for v in out]


def _rnn(sd, prefix, rnn_type, seq):
    """nn.GRU / nn.LSTM(num_layers=1, batch_first=True) evaluated from raw weights (layers.py:27-30)."""
    w_ih, w_hh = sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"]
    b_ih, b_hh = sd.get(prefix + "bias_ih_l0"), sd.get(prefix + "bias_hh_l0")
    hid = w_hh.shape[1]
    mod = (torch.nn.LSTM if rnn_type == "LSTM" else torch.nn.GRU)(w_ih.shape[1], hid, 1, bias=b_ih is not None,
                                                                 batch_first=True).to(w_ih.dtype)   # float64 runs = exact truth
    with torch.no_grad():
        mod.weight_ih_l0.copy_(w_ih)
        mod.weight_hh_l0.copy_(w_hh)
        if b_ih is not None:
            mod.bias_ih_l0.copy_(b_ih)
            mod.bias_hh_l0.copy_(b_hh)
    for p in mod.parameters():
        p.requires_grad_(False)
    return mod(seq)[0]


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
