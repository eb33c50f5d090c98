"""r5: the DiT-L/2 GEMM shapes on the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16 in / bf16 out, no epilogue) beside
ln3d_gemm_bf16 (plain bf16 epilogue and the fused epilogue the model uses), isolated loops of 50 launches, HIP events.  A reference
point for `roofline.frac`, not a product path (torch's GEMM is never called by the package)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ln3diff_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
shapes = [('qkv', 12288, 3072, 1024), ('proj', 12288, 1024, 1024), ('fc1', 12288, 4096, 1024), ('fc2', 12288, 1024, 4096),
          ('fc1 i23d', 49152, 4096, 1024), ('fc2 i23d', 49152, 1024, 4096)]


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


print('| GEMM | M x N x K | torch.matmul us (TFLOP/s) | ln3d plain bf16 us (TFLOP/s) | ln3d fused epilogue us |')
print('|---|---|---|---|---|')
for name, M, N, K in shapes:
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev) * 0.1
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    wt = w.t()
    t_blas = timeit(lambda: torch.matmul(x, wt, out=y))
    y2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_mine = timeit(lambda: ops.gemm(x, w, b, ops.EPI_BF16, y2))
    err = float(((y2.float() - (y.float() + b)).norm() / (y.float() + b).norm()))
    if name.startswith('fc1'):
        t_f = timeit(lambda: ops.gemm(x, w, b, ops.EPI_GELU_ERF, y2))
    elif name.startswith('fc2') or name == 'proj':
        xt = torch.zeros(M, N, device=dev)
        g = torch.randn(1, N, device=dev)
        t_f = timeit(lambda: ops.gemm(x, w, b, ops.EPI_GATE_RES, xt, gate=g, gate_rows=768, gate_ld=0))
    else:
        t_f = float('nan')
    fl = 2.0 * M * N * K
    print('| %s | %d x %d x %d | %.1f (%.0f) | %.1f (%.0f) | %.1f | rel diff %.1e' % (name, M, N, K, t_blas, fl / t_blas / 1e6, t_mine, fl / t_mine / 1e6, t_f, err))
