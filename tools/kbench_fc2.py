#!/usr/bin/env python
"""fc2 (M = 12288 / 49152, N = 1024, K = 4096, gate / residual epilogue) per forced tile configuration (LN3D_GEMM_TILE is read once per
process: one process per setting)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops
from kbench_half import timeit
dev = 'cuda'
tag = os.environ.get('LN3D_GEMM_TILE', 'default')
for M in (12288, 49152):
    N, K = 1024, 4096
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16); b = torch.randn(N, device=dev) * 0.02
    out = torch.randn(M, N, device=dev); gate = torch.randn(M // 768, 6 * N, device=dev)
    t1 = timeit(lambda: ops.gemm(x, w, b, ops.EPI_GATE_RES, out, None, gate=gate, gate_rows=768, gate_ld=6 * N))
    print(f'tile {tag:8s} fc2 M {M:6d} N 1024 K 4096 GATE_RES {t1:6.1f} us  {2.0 * M * N * K / t1 / 1e6:7.1f} TF/s')
