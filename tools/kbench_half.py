#!/usr/bin/env python
"""GEMM timings at HALF the network batch (the conditional half of a CFG batch: M = 8 x 768 = 6144): which tile configuration
serves the cross-attention GEMMs when the unconditional half is folded away (LN3D_GEMM_TILE read once per process: one process per
setting)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops
dev = 'cuda'
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
if __name__ == '__main__':
    tag = os.environ.get('LN3D_GEMM_TILE', 'default')
    for M in (6144, 12288):
        N = K = 1024
        x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16); b = torch.randn(N, device=dev) * 0.02
        out = torch.randn(M, N, device=dev); gate = torch.randn(M // 768, 6 * N, device=dev)
        ob = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t1 = timeit(lambda: ops.gemm(x, w, b, ops.EPI_GATE_RES, out, None, gate=gate, gate_rows=768, gate_ld=6 * N))
        t2 = timeit(lambda: ops.gemm(x, w, None, ops.EPI_BF16, ob))
        print(f'tile {tag:8s} M {M:6d} N 1024 K 1024: GATE_RES {t1:6.1f} us   plain bf16 {t2:6.1f} us')

