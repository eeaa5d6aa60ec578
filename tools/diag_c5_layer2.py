"""Where does the common 2.65e-2 dX error of tools/diag_c5_layer.py come from?  Finds the neighbour the bad rows share and compares dH there."""
import os, sys, time
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from ctgcn_amd import ops
from ctgcn_amd.helper import core_adj_from_scipy
from ctgcn_amd.layers import CoreDiffusion, rnn_reduce_norm
from ctgcn_amd.synth import snapshot_rows
from oracle import oracle as O, torch_path as TP

n, edges = 1_000_000, 8_000_000
DEV = "cuda:0"
u, v, picks = snapshot_rows(n, edges, 16, cumulative=True)
uu, vv = u[picks[15]], v[picks[15]]
g = sp.coo_matrix((np.ones(2 * len(uu)), (np.concatenate([uu, vv]), np.concatenate([vv, uu]))), shape=(n, n)).tocsr(); g.sort_indices()
core = np.minimum(O.core_numbers(g), 8)
adj, _, _ = core_adj_from_scipy(g, 8, DEV)
mats = O.core_adj_list([O.kcore_matrices(g, core)], 0, 1, 1, max_core=8)[0]
deg = np.diff(g.indptr)
rng = np.random.default_rng(11)
iso = np.flatnonzero(deg == 0)
rows = np.unique(np.concatenate([rng.choice(n, 131072, replace=False), np.argsort(-deg, kind="stable")[:256], rng.choice(iso, min(4096, len(iso)), replace=False)]))
torch.manual_seed(5)
layer = CoreDiffusion(128, 128)
with torch.no_grad():
    layer.norm.weight.uniform_(0.5, 1.5); layer.norm.bias.uniform_(-0.5, 0.5)
x = torch.randn(n, 128); Gs = torch.randn(len(rows), 128)

# fp64 truth with the stacked H as a leaf we can read the gradient of
sd = {"l." + k: p.detach().double().clone().requires_grad_(True) for k, p in layer.state_dict().items() if not k.startswith("linear.")}
xd = x.detach().clone().double().requires_grad_(True)
hs = TP.aggregate_loop(TP._rows_of(mats, rows, torch.float64), xd)
seq = torch.stack(hs, 0).transpose(0, 1)
seq.retain_grad()
pre = TP._rnn_grad(sd, "l.rnn.", "GRU", seq).sum(1)
pre.retain_grad()
out = F.layer_norm(pre, (128,), sd["l.norm.weight"], sd["l.norm.bias"])
(out * Gs.double()).sum().backward()
dx64, dH64, dpre64 = xd.grad, seq.grad, pre.grad

os.environ["CTGCN_TRAIN_FUSED"] = "0"
import copy
L = copy.deepcopy(layer).to(DEV)
xg = x.detach().clone().to(DEV).requires_grad_(True)
G = torch.zeros(n, 128, device=DEV); sel = torch.from_numpy(rows).to(DEV); G[sel] = Gs.to(DEV)
H = L.aggregate(xg, adj); H.retain_grad()
o = rnn_reduce_norm(L.rnn, L.norm, H, reduce_sum=True)
(o * G).sum().backward()
dx = xg.grad.cpu().double(); dH = H.grad[sel].cpu().double()
E = (dx - dx64).abs().max(1).values
bad = torch.nonzero(E > 0.5 * E.max()).flatten().numpy()
print("rows with error > half the max: %d, max %.3e" % (len(bad), E.max()))
cnt = {}
for r in bad[:2000]:
    for c in g.indices[g.indptr[r]:g.indptr[r + 1]]:
        cnt[c] = cnt.get(c, 0) + 1
top = sorted(cnt.items(), key=lambda kv: -kv[1])[:5]
print("most shared neighbours:", [(c, k, int(deg[c]), int(core[c]), int(c in set(rows.tolist()))) for c, k in top])
edH = (dH - dH64).abs()
w = int(edH.reshape(len(rows), -1).max(1).values.argmax())
print("worst dH row: node %d deg %d core %d: max |dH err| %.3e, max |dH| there %.3e; per slot err %s" % (
    rows[w], deg[rows[w]], core[rows[w]], edH[w].max(), dH64[w].abs().max(), ["%.1e" % e for e in edH[w].max(1).values.tolist()]))
print("   |H| max at that row %.3e; |pre-norm| max %.3e; dpre64 max %.3e" % (seq[w].abs().max(), pre[w].abs().max(), dpre64[w].abs().max()))
c = top[0][0]
if c in set(rows.tolist()):
    i = int(np.searchsorted(rows, c))
    print("shared neighbour %d: dH err per slot %s, |dH| per slot %s" % (c, ["%.1e" % e for e in edH[i].max(1).values.tolist()], ["%.1e" % e for e in dH64[i].abs().max(1).values.tolist()]))
    print("   pre-norm row: mean %.4e var %.4e; |H| max %.3e" % (pre[i].mean(), pre[i].var(unbiased=False), seq[i].abs().max()))
rel_all = edH.reshape(len(rows), -1).max(1).values / dH64.abs().reshape(len(rows), -1).max(1).values.clamp_min(1e-30)
print("dH relative error per row: median %.2e, 99.9%% %.2e, max %.2e" % (rel_all.median(), rel_all.quantile(0.999), rel_all.max()))
