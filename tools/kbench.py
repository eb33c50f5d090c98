#!/usr/bin/env python
"""Micro-benchmarks of the HIP kernels at the DiT-L/2 (B=8, CFG -> 16 x 768 tokens) shapes. GPU box only."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops  # noqa: E402

dev = 'cuda'


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


def gemm_case(name, M, N, K, epi, **kw):
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.02
    if epi == ops.EPI_GATE_RES:
        out = torch.randn(M, N, device=dev)
        gate = torch.randn(M // 768, 6 * N, device=dev)
        cp = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if kw.get('copy') else None
        f = lambda: ops.gemm(x, w, b, epi, out, cp, gate=gate, gate_rows=768, gate_ld=6 * N)
    elif epi == ops.EPI_HEADS:
        H = 16
        q = torch.zeros(M // 768, H, 768, 64, device=dev, dtype=torch.bfloat16)
        k = torch.zeros_like(q)
        vt = torch.zeros(M // 768, H, 64, 768, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm(x, w, b, epi, q, k, vt, M=M, tokens=768, tok_pad=768, heads=H, head_dim=64, transpose_mask=kw.get('tmask', 0b100))
    elif epi == ops.EPI_F32:
        out = torch.empty(M, N, device=dev)
        f = lambda: ops.gemm(x, w, b, epi, out)
    else:
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        f = lambda: ops.gemm(x, w, b, epi, out)
    us = timeit(f)
    print(f'{name:34s} M{M} N{N} K{K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s')


M = 16 * 768
gemm_case('qkv (HEADS, V^T)', M, 3072, 1024, ops.EPI_HEADS)
gemm_case('qkv shape, plain BF16', M, 3072, 1024, ops.EPI_BF16)
gemm_case('qkv (HEADS, no transpose)', M, 3072, 1024, ops.EPI_HEADS, tmask=0)
gemm_case('proj (GATE_RES)', M, 1024, 1024, ops.EPI_GATE_RES)
gemm_case('proj (GATE_RES + bf16 copy)', M, 1024, 1024, ops.EPI_GATE_RES, copy=True)
gemm_case('cross to_q (BF16 plain)', M, 1024, 1024, ops.EPI_BF16)
gemm_case('fc1 (GELU_ERF)', M, 4096, 1024, ops.EPI_GELU_ERF)
gemm_case('fc1 shape, plain BF16', M, 4096, 1024, ops.EPI_BF16)
gemm_case('fc2 (GATE_RES)', M, 1024, 4096, ops.EPI_GATE_RES)
gemm_case('fc2 shape, plain BF16', M, 1024, 4096, ops.EPI_BF16)
gemm_case('big square plain BF16', 8192, 8192, 8192, ops.EPI_BF16)
gemm_case('adaLN all layers (F32, M=16)', 16, 24 * 6144 + 2048, 1024, ops.EPI_F32)

B, H, N, Dh = 16, 16, 768, 64
q = torch.randn(B, H, N, Dh, device=dev).to(torch.bfloat16)
k = torch.randn(B, H, N, Dh, device=dev).to(torch.bfloat16)
vt = torch.randn(B, H, Dh, N, device=dev).to(torch.bfloat16)
o = torch.empty(B, N, H * Dh, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: ops.attention(q, k, vt, o, B, H, N, N, N, N, Dh))
print(f'self-attn 16x16x768x768x64       : {us:8.1f} us  {4.0 * N * N * H * Dh * B / us / 1e6:7.1f} TF/s')
kc = torch.randn(B, H, 128, Dh, device=dev).to(torch.bfloat16)
vc = torch.randn(B, H, Dh, 128, device=dev).to(torch.bfloat16)
us = timeit(lambda: ops.attention(q, kc, vc, o, B, H, N, N, 77, 128, Dh))
print(f'cross-attn 768x77                 : {us:8.1f} us  {4.0 * N * 77 * H * Dh * B / us / 1e6:7.1f} TF/s')

x = torch.randn(M, 1024, device=dev)
y = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
mod = torch.randn(16, 6 * 1024, device=dev)
us = timeit(lambda: ops.norm_modulate(x, y, M, 1024, shift=mod, scale=mod[:, 1024:], mod_rows=768, mod_ld=6144))
print(f'LN+modulate 12288x1024            : {us:8.1f} us  {M * 1024 * 6 / us / 1e3:7.1f} GB/s')

# yardstick only (not a product path): the vendor library GEMM at the same shapes
if os.environ.get('KBENCH_VENDOR', '1') == '1':
    for (nm, M_, N_, K_) in [('vendor fc1 shape', M, 4096, 1024), ('vendor fc2 shape', M, 1024, 4096),
                             ('vendor qkv shape', M, 3072, 1024), ('vendor proj shape', M, 1024, 1024),
                             ('vendor big square', 8192, 8192, 8192)]:
        a = torch.randn(M_, K_, device=dev).to(torch.bfloat16)
        w = (torch.randn(N_, K_, device=dev) * 0.03).to(torch.bfloat16)
        us = timeit(lambda: torch.matmul(a, w.t()))
        print(f'{nm:34s} M{M_} N{N_} K{K_}: {us:8.1f} us  {2.0 * M_ * N_ * K_ / us / 1e6:7.1f} TF/s')

# fused query projection + cross-attention (LN3D_EPI_CROSS_ATTN) vs the two separate kernels above
Lc, lpad = 77, 128
xq = torch.randn(M, 1024, device=dev).to(torch.bfloat16)
wq = (torch.randn(1024, 1024, device=dev) * 0.03).to(torch.bfloat16)
kcx = torch.zeros(16, 16, lpad, 64, device=dev, dtype=torch.bfloat16); kcx[:, :, :Lc] = torch.randn(16, 16, Lc, 64, device=dev).to(torch.bfloat16)
vtx = torch.randn(16, 16, 64, lpad, device=dev).to(torch.bfloat16)
oc = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
us = timeit(lambda: ops.gemm(xq, wq, None, ops.EPI_CROSS_ATTN, oc, kcx, vtx, M=M, tokens=768, heads=16, head_dim=64, ctx_keys=Lc,
                             ctx_pad=lpad, ctx_scale=0.125))
print(f'to_q + cross-attn fused (CROSS_ATTN epilogue)              : {us:8.1f} us')
