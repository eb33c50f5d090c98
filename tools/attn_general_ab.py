#!/usr/bin/env python
"""Timing + fp32 check of ln3d_attention_bf16 on the shapes that take the GENERAL ring kernel (attn_kernel<DH, OCC, DT>): DiT-XL/2's 72-wide heads
stored 128 wide, full 128-wide heads, Dh 64 with a ragged key count.  LN3D_LIB selects the library build (same-box A/B)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LN3D_LIB'):
    from ln3diff_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['LN3D_LIB'])
from ln3diff_amd import ops          # noqa: E402

dev = 'cuda'
torch.manual_seed(0)
# (name, B, H, Nq, Nk, Dh stored, Dh true)
CASES = [('xl2 72-in-128', 16, 16, 768, 768, 128, 72), ('xl2 72-in-80', 16, 16, 768, 768, 80, 72), ('heads 80-in-80', 8, 16, 1024, 1024, 80, 80),
         ('80-in-80 ragged', 3, 5, 333, 1000, 80, 80), ('dh128', 8, 16, 768, 768, 128, 128), ('dh64 ragged 257', 32, 16, 257, 257, 64, 64),
         ('dh64 ragged 1000', 4, 16, 700, 1000, 64, 64), ('dh64 short 77', 16, 16, 768, 77, 64, 64)]
for name, B, H, Nq, Nk, Dh, Dt in CASES:
    Nqp, Nkp = (Nq + 63) // 64 * 64, (Nk + 63) // 64 * 64
    q = torch.zeros(B, H, Nqp, Dh, device=dev); k = torch.zeros(B, H, Nkp, Dh, device=dev); v = torch.zeros(B, H, Nkp, Dh, device=dev)
    q[:, :, :Nq, :Dt] = torch.randn(B, H, Nq, Dt, device=dev); k[:, :, :Nk, :Dt] = torch.randn(B, H, Nk, Dt, device=dev)
    v[:, :, :Nk, :Dt] = torch.randn(B, H, Nk, Dt, device=dev)
    k[:, 1::3, 5] *= 6.0                                     # a spiked key in some heads: the deferred-rebase branch runs
    qb, kb, vb = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    # V^T in the ABI's key order: keys of every 16-group permuted to [0-3, 8-11, 4-7, 12-15]
    idx = torch.arange(Nkp, device=dev)
    perm = (idx & ~12) | ((idx & 4) << 1) | ((idx & 8) >> 1)
    vt = torch.empty(B, H, Dh, Nkp, device=dev, dtype=torch.bfloat16)
    vt[:, :, :, perm] = vb.transpose(2, 3)
    vt = vt.contiguous()
    o = torch.zeros(B, Nq, H * Dt, device=dev, dtype=torch.bfloat16)
    scale = Dt ** -0.5
    f = lambda: ops.attention(qb, kb, vt, o, B, H, Nq, Nqp, Nk, Nkp, Dh, scale=scale, dh_true=(Dt if Dt != Dh else 0))
    try:
        f()
    except Exception as e:
        print(f'{name:18s} not supported by this build ({str(e)[:60]})')
        continue
    ref = torch.softmax((qb.float()[:, :, :Nq] @ kb.float()[:, :, :Nk].transpose(2, 3)) * scale, -1) @ vb.float()[:, :, :Nk]
    ref = ref[..., :Dt].permute(0, 2, 1, 3).reshape(B, Nq, H * Dt)
    err = float((o.float() - ref).norm() / ref.norm())
    o2 = o.clone(); f()
    same = bool(torch.equal(o, o2))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(50):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50)
    fl = 4.0 * B * H * Nq * Nk * Dt
    print(f'{name:18s} B*H {B * H:4d} Nq {Nq:5d} Nk {Nk:5d}: {best * 1e3:7.1f} us  {fl / best / 1e9:7.1f} TFLOP/s  rel-L2 vs fp32 {err:.2e}  repeat-identical {same}')
