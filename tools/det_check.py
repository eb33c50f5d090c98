#!/usr/bin/env python
"""Bitwise run-to-run determinism of the GEMM epilogues / norm kernels at the DiT-L/2 shapes (GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ln3diff_amd import ops
dev = 'cuda'
M = 16 * 768
def chk(name, f, outs):
    f(); torch.cuda.synchronize(); ref = [o.clone() for o in outs()]
    bad = 0
    for _ in range(4):
        f(); torch.cuda.synchronize()
        bad += sum(int((a.view(torch.int16 if a.dtype == torch.bfloat16 else torch.int32) != b.view(torch.int16 if b.dtype == torch.bfloat16 else torch.int32)).sum()) for a, b in zip(outs(), ref))
    print(f'{name:28s} differing elements over 4 repeats: {bad}')
x = torch.randn(M, 1024, device=dev).to(torch.bfloat16); w = (torch.randn(4096, 1024, device=dev) * 0.03).to(torch.bfloat16); b = torch.randn(4096, device=dev) * 0.02
y = torch.empty(M, 4096, device=dev, dtype=torch.bfloat16)
chk('fc1 GELU_ERF', lambda: ops.gemm(x, w, b, ops.EPI_GELU_ERF, y), lambda: [y])
w2 = (torch.randn(1024, 4096, device=dev) * 0.03).to(torch.bfloat16); b2 = torch.randn(1024, device=dev) * 0.02
res0 = torch.randn(M, 1024, device=dev); res = res0.clone(); gate = torch.randn(16, 6144, device=dev); cp = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
def f2():
    res.copy_(res0); ops.gemm(y, w2, b2, ops.EPI_GATE_RES, res, cp, gate=gate, gate_rows=768, gate_ld=6144)
chk('fc2 GATE_RES', f2, lambda: [res, cp])
wq = (torch.randn(3072, 1024, device=dev) * 0.03).to(torch.bfloat16); bq = torch.randn(3072, device=dev) * 0.02
q = torch.zeros(16, 16, 768, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros(16, 16, 64, 768, device=dev, dtype=torch.bfloat16)
chk('qkv HEADS', lambda: ops.gemm(x, wq, bq, ops.EPI_HEADS, q, k, vt, M=M, tokens=768, tok_pad=768, heads=16, head_dim=64, transpose_mask=0b100), lambda: [q, k, vt])
o = torch.empty(16, 768, 1024, device=dev, dtype=torch.bfloat16)
chk('self-attention', lambda: ops.attention(q, k, vt, o, 16, 16, 768, 768, 768, 768, 64), lambda: [o])
xf = torch.randn(M, 1024, device=dev); yn = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16); mod = torch.randn(16, 6144, device=dev)
chk('LN+modulate', lambda: ops.norm_modulate(xf, yn, M, 1024, shift=mod, scale=mod[:, 1024:], mod_rows=768, mod_ld=6144), lambda: [yn])
Lc, lpad = 77, 128
kcx = torch.zeros(16, 16, lpad, 64, device=dev, dtype=torch.bfloat16); kcx[:, :, :Lc] = torch.randn(16, 16, Lc, 64, device=dev).to(torch.bfloat16)
vtx = torch.randn(16, 16, 64, lpad, device=dev).to(torch.bfloat16); wq1 = (torch.randn(1024, 1024, device=dev) * 0.03).to(torch.bfloat16); oc = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
chk('to_q + cross-attn', lambda: ops.gemm(x, wq1, None, ops.EPI_CROSS_ATTN, oc, kcx, vtx, M=M, tokens=768, heads=16, head_dim=64, ctx_keys=Lc, ctx_pad=lpad, ctx_scale=0.125), lambda: [oc])
# VAE decode (GroupNorm statistics are reduced in a fixed order since r2; fp32 atomics there made the planes differ by 3e-2)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_decode_gpu import build_decoder
from conftest import load_synth
from ln3diff_amd.synth import synth_input
dec = build_decoder(128, 2, 2); load_synth(dec, 0); dec = dec.cuda()
lat = {'latent_normalized_2Ddiffusion': synth_input('latent', (2, 12, 32, 32), 5).cuda()}
outs = []
def fdec():
    outs[:] = [dec.vit_decode_postprocess(dec.vit_decode_backbone(lat, 128), {})['latent_after_vit']]
chk('VAE decode (tiny)', fdec, lambda: outs)
