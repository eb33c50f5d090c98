#!/usr/bin/env python
"""GEMM / norm micro-benchmarks at the I23D configs[2] shapes (network batch 64 x 768 tokens + 256 appended). GPU box only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


dev = 'cuda'
Bn, N, NA, D, H = 64, 768, 1024, 1024, 16
M, MA = Bn * N, Bn * NA
x = torch.randn(MA, D, device=dev).to(torch.bfloat16)
wq = (torch.randn(3 * D, D, device=dev) * 0.03).to(torch.bfloat16); bq = torch.randn(3 * D, device=dev) * 0.02
q = torch.zeros(Bn, H, NA, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(Bn, H, 64, NA, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: ops.gemm(x, wq, bq, ops.EPI_HEADS, q, k, vt, M=MA, tokens=NA, tok_pad=NA, heads=H, head_dim=64, transpose_mask=0b100))
print(f'qkv HEADS  M{MA} N3072 K1024: {us:8.1f} us  {2.0 * MA * 3072 * 1024 / us / 1e6:7.1f} TF/s')
wn = torch.ones(64, device=dev)
us = timeit(lambda: ops.rmsnorm_heads(q, wn, Bn * H * NA, 64))
print(f'rmsnorm_heads on q [{Bn},{H},{NA},64]: {us:8.1f} us')
xc = torch.randn(M, D, device=dev).to(torch.bfloat16)
wc = (torch.randn(D, D, device=dev) * 0.03).to(torch.bfloat16)
qc = torch.zeros(Bn, H, N, 64, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: ops.gemm(xc, wc, None, ops.EPI_HEADS, qc, M=M, tokens=N, tok_pad=N, heads=H, head_dim=64))
print(f'cross q HEADS M{M} N1024 K1024: {us:8.1f} us  {2.0 * M * 1024 * 1024 / us / 1e6:7.1f} TF/s')
