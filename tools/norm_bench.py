"""norm_modulate at the two workloads' shapes (GPU box): LayerNorm + modulate 12288 x 1024 (T23D), RMSNorm + table modulate with
appended rows 49152 x 1024 (I23D).  LN3D_LIB selects an alternative build of the library."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LN3D_LIB'):
    from ln3diff_amd import _lib
    _lib.LIB_PATH = os.environ['LN3D_LIB']
from ln3diff_amd import ops
dev = 'cuda'
torch.manual_seed(0)


def timeit(f, n=200):
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (Bn, N, NA, kind) in ((16, 768, 768, 0), (64, 768, 1024, 1)):
    M, D = Bn * N, 1024
    x = torch.randn(M, D, device=dev)
    y = torch.empty(Bn * NA, D, device=dev, dtype=torch.bfloat16)
    mod = torch.randn(Bn, 6 * D, device=dev) * 0.1
    w = torch.randn(D, device=dev) if kind else None
    f = lambda: ops.norm_modulate(x, y, M, D, kind=kind, eps=1e-6, weight=w, shift=mod, scale=mod[:, D:], mod_rows=N, mod_ld=6 * D,
                                  rows_in=N, rows_out=NA)
    us = timeit(f)
    # in a real layer x was just written by the previous GEMM's epilogue: interleave a writer of x so that it is as warm as in situ
    print("norm_modulate kind %d  %6d x %d -> %6d rows: %6.2f us  (%.2f TB/s of x read + y written)" % (kind, M, D, Bn * NA, us, (M * D * 6) / us / 1e6))
