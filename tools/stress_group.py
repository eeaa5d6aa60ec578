#!/usr/bin/env python3
"""Stress: repeated small-window inference forwards must be bit-identical (grouped / per-snapshot launches, any number of streams).
  python tools/stress_group.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctgcn_amd import CTGCN  # noqa: E402
from ctgcn_amd.helper import core_adj_from_scipy  # noqa: E402
from ctgcn_amd.synth import dynamic_graph  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
n, T = 3001, 5
graphs = dynamic_graph(n, avg_deg=6, snapshots=T, seed=3)
adjs = [core_adj_from_scipy(g, 6, dev)[0] for g in graphs]
for dedup in ("0", "1"):
    os.environ["CTGCN_DEDUP"] = dedup
    for kw, feat, hid in ((dict(trans_num=1, diffusion_num=2, model_type="C", trans_activate_type="L"), 40, 64),
                          (dict(trans_num=1, diffusion_num=2, model_type="C", trans_activate_type="L"), 20, 128)):
        torch.manual_seed(0)
        model = CTGCN(feat, hid, 128, duration=T, **kw).to(dev).eval()
        xs = [torch.randn(n, feat, device=dev) for _ in range(T)]
        ref = None
        bad = 0
        for it in range(iters):
            os.environ["CTGCN_GROUP"] = "1" if it % 2 == 0 else "0"
            junk = torch.empty((it * 7919) % 50_000_000 + 1, device=dev)      # perturb the allocator
            with torch.no_grad():
                out = model(xs, adjs)
            del junk
            if ref is None:
                ref = out.clone()
            elif not torch.equal(out, ref):
                bad += 1
                d = (out != ref)
                rows = d.any(-1).nonzero()
                print("  mismatch it=%d group=%s: %d elements, first (t, node) %s, max |diff| %.3e" % (
                    it, os.environ["CTGCN_GROUP"], int(d.sum()), rows[0].tolist() if rows.numel() else None, float((out - ref).abs().max())), flush=True)
        print("dedup=%s hid=%d streams=%s: %d mismatches in %d forwards" % (dedup, hid, os.environ.get("CTGCN_STREAMS", "default"), bad, iters), flush=True)
