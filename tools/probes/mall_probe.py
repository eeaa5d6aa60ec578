#!/usr/bin/env python
"""Probe: does a buffer that was just WRITTEN get read back from the 256 MB Infinity Cache (MALL) instead of HBM?
Reads `size` MB right after writing it (warm) vs after 2 GB of unrelated traffic (cold)."""
import torch

dev = "cuda:0"
flush = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)      # 2 GB


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)


for mb in (32, 64, 128, 192, 256, 512):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    res = {}
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(5):
            a.fill_(1.0)                       # write the buffer
            if mode == "cold":
                flush.add_(1.0)                # 4 GB of other traffic
            torch.cuda.synchronize()
            ts.append(timed(lambda: a.sum()))  # read it back
        res[mode] = min(ts)
    print("%4d MB: read after write %.3f ms = %.2f TB/s | read cold %.3f ms = %.2f TB/s" %
          (mb, res["warm"], mb / 1024 / 1024 * 1e3 / res["warm"] * 1.048576, res["cold"], mb / 1024 / 1024 * 1e3 / res["cold"] * 1.048576))
