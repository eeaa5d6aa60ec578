"""How many forwards the grouped launches' descriptor-table cache (ops._GroupTables, ABI 28) needs to settle: tables written / found current and
cache misses per call site after every forward of a small one-hot window.  python tools/table_cache_probe.py"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, sys.path[0] + "/tests")
from ctgcn_amd import CTGCN, _lib, ops
from test_gpu_group import _window, DEV
lib = _lib.load()
n, T = 3001, 6
adjs = _window(n, T, 6, 6, seed=3)
torch.manual_seed(0)
model = CTGCN(n, 500, 128, 1, 2, T).to(DEV).eval()
idx = torch.arange(n, device=DEV).repeat(2, 1)
xs = [torch.sparse_coo_tensor(idx, torch.ones(n, device=DEV), (n, n)) for _ in range(T)]
with torch.no_grad():
    for i in range(12):
        out = model(xs, adjs)
        torch.cuda.synchronize()
        print(i, "written", int(lib.ctgcn_table_uploads(0)), "current", int(lib.ctgcn_table_uploads(1)), dict(ops._group_tables.misses), len(ops._group_tables.entries))
