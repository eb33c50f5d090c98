"""Ablations of the ring GEMM (bench-only LN3D_GEMM_ABL bits: 1 skip epilogue, 2 two K-stages only, 4 no DMA in steady state)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops
from kbench import timeit
dev = 'cuda'
NAMES = {0: 'full', 1: 'no epilogue', 4: 'no DMA', 5: 'no DMA, no epilogue', 2: 'prologue + 2 stages + epilogue', 3: 'prologue + 2 stages'}
for (M, N, K) in [(12288, 4096, 1024), (12288, 1024, 4096), (12288, 1024, 1024), (8192, 8192, 8192)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for abl, nm in NAMES.items():
        os.environ['LN3D_GEMM_ABL'] = str(abl)
        ops.reload_env()          # the switches are parsed once per process
        us = timeit(lambda: ops.gemm(x, w, None, ops.EPI_BF16, out))
        print(f'M{M} N{N} K{K} ABL={abl} ({nm}): {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s-equiv')
os.environ.pop('LN3D_GEMM_ABL', None)
ops.reload_env()
