"""Diagnostic for tests/test_gpu_configs.py::test_config5_training_layer_gradients_at_full_size: where do forward / dX errors sit?
python tools/diag_c5_layer.py [n_nodes] [edges]   (default: config 5's snapshot 15)"""
import os, sys, time
import numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import ops
from ctgcn_amd.helper import core_adj_from_scipy
from ctgcn_amd.layers import CoreDiffusion
from ctgcn_amd.synth import snapshot_rows
from oracle import oracle as O, torch_path as TP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
edges = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
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
rows = np.unique(np.concatenate([rng.choice(n, min(n, 131072), replace=False), np.argsort(-deg, kind="stable")[:256], rng.choice(iso, min(4096, len(iso)), replace=False)]))
torch.manual_seed(5)
layer = CoreDiffusion(128, 128)
with torch.no_grad():
    layer.norm.weight.uniform_(0.5, 1.5); layer.norm.bias.uniform_(-0.5, 0.5)
x = torch.randn(n, 128); Gs = torch.randn(len(rows), 128)

def truth(dtype):
    sd = {"l." + k: p.detach().to(dtype).clone().requires_grad_(True) for k, p in layer.state_dict().items() if not k.startswith("linear.")}
    xd = x.detach().clone().to(dtype).requires_grad_(True)
    saved = TP._rnn; TP._rnn = TP._rnn_grad
    try:
        out = TP.core_diffusion(sd, "l.", xd, TP._rows_of(mats, rows, dtype))
    finally:
        TP._rnn = saved
    (out * Gs.to(dtype)).sum().backward()
    return out.detach(), xd.grad, {k[2:]: p.grad for k, p in sd.items()}

t0 = time.time(); o64, dx64, g64 = truth(torch.float64); print("fp64 truth %.1fs" % (time.time() - t0))
o32, dx32, g32 = truth(torch.float32)

def hip(fused):
    import copy
    os.environ["CTGCN_TRAIN_FUSED"] = "1" if fused else "0"
    L = copy.deepcopy(layer).to(DEV)
    xg = x.detach().clone().to(DEV).requires_grad_(True)
    G = torch.zeros(n, 128, device=DEV); G[torch.from_numpy(rows).to(DEV)] = Gs.to(DEV)
    out = L(xg, adj); (out * G).sum().backward()
    return out.detach()[torch.from_numpy(rows).to(DEV)].cpu(), xg.grad.cpu(), {k: p.grad.cpu() for k, p in L.named_parameters() if p.grad is not None}

def report(tag, o, dx, gr):
    eo = (o.double() - o64).abs(); edx = (dx.double() - dx64).abs()
    print("%s: forward max err %.2e (row deg %d); dX max err %.2e rel-to-max %.2e (max |dX| %.3e)" % (
        tag, eo.max(), deg[rows[int(eo.max(1).values.argmax())]], edx.max(), edx.max() / dx64.abs().max(), dx64.abs().max()))
    worst = torch.topk(edx.max(1).values, 8).indices.numpy()
    for r in worst:
        print("   row %8d deg %6d core %d in_sample %d  |dX|max %.3e err %.3e" % (r, deg[r], core[r], int(r in set(rows.tolist())), dx64[r].abs().max(), edx[r].max()))
    for k in gr:
        if k in g64:
            print("   %-20s rel %.2e" % (k, (gr[k].double() - g64[k]).abs().max() / g64[k].abs().max()))

report("fp32 CPU autograd", o32, dx32, g32)
report("HIP fused", *hip(True))
report("HIP round-3 path", *hip(False))
