#!/usr/bin/env python3
"""What does the library's fp16 GEMM with fp32 output reach on the shapes of the split GEMM?  (torch.mm(..., out_dtype=torch.float32) -> hipBLASLt.)
K' = 3 Kp: the three products of the fp16 x 2 split as ONE GEMM over concatenated planes [A1 | A2 | A1] x [B2 | B1 | B1]^T."""
import torch
dev = torch.device("cuda:0")
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
for name, M, N, K in [("enron proj", 435180, 384, 512), ("fb mlp0", 60730, 500, 1792), ("fb mlp1", 60730, 500, 512), ("fb mlp2", 60730, 128, 512), ("math proj", 197920, 384, 512)]:
    for mult in (1, 3):
        a = torch.randn(M, K * mult, device=dev).half()
        b = torch.randn(N, K * mult, device=dev).half()
        out = torch.empty(M, N, device=dev, dtype=torch.float32)
        try:
            ms = t(lambda: torch.mm(a, b.t(), out_dtype=torch.float32))
            how = "mm out_dtype=f32"
        except Exception as ex:
            print(name, "mm out_dtype failed:", str(ex)[:120])
            ms = t(lambda: torch.mm(a, b.t()))
            how = "mm f16 out"
        fl = 2.0 * M * N * K * mult
        print("%-12s M=%d N=%d K=%d  %s: %.3f ms = %.0f TF/s executed (%.0f TF/s fp32-equivalent if K'=3K)" % (name, M, N, K * mult, how, ms, fl / ms / 1e9, 2.0 * M * N * K / ms / 1e9), flush=True)
    del a, b, out
