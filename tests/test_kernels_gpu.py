"""GPU parity of the individual HIP kernels (through the C ABI) against plain torch fp32 / the oracle.
Tolerances: bf16 GEMM/attention operands -> rel-L2 <= 1e-2 vs an fp32 reference computed from the SAME
bf16-rounded operands (so the bound measures the kernel, not the rounding of the inputs): <= 2e-3."""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "small", "x7", "x8", "x9", "x12", "x13", "x14", "x16"])
def ops(hip_lib, request):
    """Every kernel test runs with the GEMM tile selection left to the library and forced to each tile config."""
    import os
    from ln3diff_amd import ops as o
    old = os.environ.get("LN3D_GEMM_TILE")
    if request.param == "auto":
        os.environ.pop("LN3D_GEMM_TILE", None)
    else:
        os.environ["LN3D_GEMM_TILE"] = {"large": "l", "small": "s"}.get(request.param, request.param)
    o.reload_env()                    # the library parses its switches once per process
    yield o
    if old is None:
        os.environ.pop("LN3D_GEMM_TILE", None)
    else:
        os.environ["LN3D_GEMM_TILE"] = old
    o.reload_env()


@pytest.fixture
def auto_tile(hip_lib):
    """For tests that do not take the tile-forcing `ops` fixture: its module-scoped parameter may still be live when they run."""
    import os
    from ln3diff_amd import ops as o
    old = os.environ.pop("LN3D_GEMM_TILE", None)
    o.reload_env()
    yield
    if old is not None:
        os.environ["LN3D_GEMM_TILE"] = old
    o.reload_env()


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 260, 192), (16, 1024, 256), (1536, 3072, 1024), (77, 512, 768),
                                   (2000, 384, 128), (1636, 132, 320)])
def test_gemm_plain_epilogues(ops, M, N, K):
    dev = 'cuda'
    g = torch.Generator().manual_seed(M + N + K)
    # ASYMMETRIC operands (transposes must show): x rows scaled by index, w cols scaled
    x = (torch.randn(M, K, generator=g) * (1 + torch.arange(M)[:, None] / M)).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05 * (1 + torch.arange(K)[None, :] / K)).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    xb, wb = _bf(x), _bf(w)
    ref = xb.float() @ wb.float().t() + b
    out = torch.empty(M, N, device=dev)
    ops.gemm(xb, wb, b, ops.EPI_F32, out)
    assert rel_l2(out, ref) < 2e-5, rel_l2(out, ref)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(xb, wb, b, ops.EPI_BF16, o16)
    assert rel_l2(o16.float(), ref) < 4e-3
    ops.gemm(xb, wb, b, ops.EPI_GELU_ERF, o16)
    assert rel_l2(o16.float(), torch.nn.functional.gelu(ref)) < 5e-3
    ops.gemm(xb, wb, b, ops.EPI_GELU_TANH, o16)
    assert rel_l2(o16.float(), torch.nn.functional.gelu(ref, approximate='tanh')) < 5e-3
    ops.gemm(xb, wb, b, ops.EPI_SILU, o16)
    assert rel_l2(o16.float(), torch.nn.functional.silu(ref)) < 5e-3
    o32 = torch.empty(M, N, device=dev)
    ops.gemm(xb, wb, b, ops.EPI_F32_SILU, o32, o16)
    assert rel_l2(o32, ref) < 2e-5 and rel_l2(o16.float(), torch.nn.functional.silu(ref)) < 5e-3


@pytest.mark.parametrize("M,N,K,bias", [(8192, 4096, 512, True), (12288, 4096, 1024, True), (4096, 2048, 1152, False), (12288, 1024, 1024, True)])
def test_gemm_persistent_tile_walk(hip_lib, auto_tile, monkeypatch, M, N, K, bias):
    """r6: the persistent one-wave-per-SIMD kernel (cfg 16) over several tiles per workgroup (2, 3), under one round (the 1.0-tile case:
    M 4096 x N 2048 = 128 tiles; N = 1024 x M 12288 = 192 tiles), without bias (zero image), K = 1152 (18 stages): bf16 and erf-GELU
    epilogues against fp32 torch on the same bf16 operands, bit-identical to the non-persistent 256 x 256 tile (same summation order),
    bit-repeatable; its output parked in registers and stored from inside the next tile's K loop must all arrive."""
    from ln3diff_amd import ops as o
    dev = 'cuda'
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * (1 + torch.arange(M)[:, None] / M)).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05 * (1 + torch.arange(K)[None, :] / K)).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    xb, wb = _bf(x), _bf(w)
    ref = xb.float() @ wb.float().t() + (b if bias else 0.0)
    outs = {}
    for tile in ('x16', 'x7'):
        monkeypatch.setenv('LN3D_GEMM_TILE', tile)
        o.reload_env()
        y = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
        o.gemm(xb, wb, b, o.EPI_BF16, y)
        yg = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
        o.gemm(xb, wb, b, o.EPI_GELU_ERF, yg)
        yg2 = torch.full((M, N), float('nan'), device=dev, dtype=torch.bfloat16)
        o.gemm(xb, wb, b, o.EPI_GELU_ERF, yg2)
        torch.cuda.synchronize()
        assert torch.isfinite(y.float()).all() and torch.isfinite(yg.float()).all()
        assert rel_l2(y.float(), ref) < 4e-3 and rel_l2(yg.float(), torch.nn.functional.gelu(ref)) < 5e-3
        assert torch.equal(yg, yg2)
        outs[tile] = (y, yg)
    monkeypatch.delenv('LN3D_GEMM_TILE')
    o.reload_env()
    assert torch.equal(outs['x16'][0], outs['x7'][0]) and torch.equal(outs['x16'][1], outs['x7'][1])


def test_gemm_identity_asymmetric(ops):
    """A = I check with an asymmetric B: catches swapped rows/cols in the accumulator write-out."""
    dev = 'cuda'
    K = 128
    x = torch.eye(K, device=dev)                                   # [M=K, K]
    w = (torch.arange(K * K, device=dev).float().reshape(K, K) % 251) / 64.0   # exactly representable
    out = torch.empty(K, K, device=dev)
    ops.gemm(_bf(x), _bf(w), None, ops.EPI_F32, out)
    assert torch.equal(out, _bf(w).float().t())


@pytest.mark.parametrize("rows_per_gate", [1, 96])
def test_gemm_gate_residual(ops, rows_per_gate):
    dev = 'cuda'
    M, N, K = 192, 256, 128
    g = torch.Generator().manual_seed(3)
    x, w, b = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.1).to(dev), torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    ng = M // rows_per_gate
    gate_full = torch.randn(ng, 3 * N, generator=g).to(dev)       # gate is a column slice of a wider matrix
    gate = gate_full[:, N:]
    xb, wb = _bf(x), _bf(w)
    ref = res + gate[:, :N].repeat_interleave(rows_per_gate, 0) * (xb.float() @ wb.float().t() + b)
    acc = res.clone()
    copy = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(xb, wb, b, ops.EPI_GATE_RES, acc, copy, gate=gate, gate_rows=rows_per_gate, gate_ld=3 * N)
    assert rel_l2(acc, ref) < 2e-5
    assert rel_l2(copy.float(), ref) < 4e-3
    acc2 = res.clone()
    ops.gemm(xb, wb, b, ops.EPI_GATE_RES, acc2)
    assert rel_l2(acc2, res + xb.float() @ wb.float().t() + b) < 2e-5


@pytest.mark.parametrize("M,N,K,rows", [(192, 256, 128, 96), (2304, 1024, 1024, 768), (6144, 1024, 1024, 768), (12288, 1024, 1024, 768),
                                         (3000, 1152, 1024, 750)])
def test_gemm_gate_residual_with_sample_rows(ops, M, N, K, rows):
    """GATE_RES with the ABI-7 per-sample row (res_bias): out0 += gate * (x W^T + b) + res_bias[sample], at the small-kernel shape,
    at the 128x192 / 4-wave tiling (under-filled launches: M = 2304, 6144), at the benchmarked 256x192 tiling (M = 12288, interior
    tiles: the double-buffered residual prefetch) and at a ragged shape whose 32-token runs cross sample boundaries."""
    dev = 'cuda'
    g = torch.Generator().manual_seed(M + N)
    x, w, b = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.05).to(dev), torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    ns = M // rows
    gate = torch.randn(ns, 2 * N, generator=g).to(dev)[:, N:]
    rb = torch.randn(ns, N, generator=g).to(dev)
    rb[ns // 2:] = 0                                                   # the conditional half carries zero rows
    xb, wb = _bf(x), _bf(w)
    lin = xb.float() @ wb.float().t() + b
    ref = res + gate.repeat_interleave(rows, 0) * lin + rb.repeat_interleave(rows, 0)
    acc = res.clone()
    copy = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(xb, wb, b, ops.EPI_GATE_RES, acc, copy, gate=gate, gate_rows=rows, gate_ld=2 * N, res_bias=rb, res_bias_ld=N)
    assert rel_l2(acc, ref) < 2e-5, rel_l2(acc, ref)
    assert rel_l2(copy.float(), ref) < 4e-3
    acc2 = res.clone()                                                 # no gate, only the rows
    ops.gemm(xb, wb, b, ops.EPI_GATE_RES, acc2, gate_rows=rows, res_bias=rb, res_bias_ld=N)
    assert rel_l2(acc2, res + lin + rb.repeat_interleave(rows, 0)) < 2e-5
    acc3, acc4 = res.clone(), res.clone()                              # bit-repeatable
    ops.gemm(xb, wb, b, ops.EPI_GATE_RES, acc3, gate=gate, gate_rows=rows, gate_ld=2 * N, res_bias=rb, res_bias_ld=N)
    ops.gemm(xb, wb, b, ops.EPI_GATE_RES, acc4, gate=gate, gate_rows=rows, gate_ld=2 * N, res_bias=rb, res_bias_ld=N)
    assert torch.equal(acc3, acc4) and torch.equal(acc3, acc)


def test_gemm_heads_split(ops):
    dev = 'cuda'
    B, T, H, Dh, K = 2, 77, 4, 64, 128
    tp = 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B * T, K, generator=g).to(dev)
    w = (torch.randn(3 * H * Dh, K, generator=g) * 0.1).to(dev)
    b = torch.randn(3 * H * Dh, generator=g).to(dev)
    xb, wb = _bf(x), _bf(w)
    ref = (xb.float() @ wb.float().t() + b).reshape(B, T, 3, H, Dh)
    q = torch.zeros(B, H, tp, Dh, device=dev, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, Dh, tp, device=dev, dtype=torch.bfloat16)
    ops.gemm(xb, wb, b, ops.EPI_HEADS, q, k, vt, M=B * T, tokens=T, tok_pad=tp, heads=H, head_dim=Dh, transpose_mask=0b100)
    assert rel_l2(q[:, :, :T].float(), ref[:, :, 0].permute(0, 2, 1, 3)) < 4e-3
    assert rel_l2(k[:, :, :T].float(), ref[:, :, 1].permute(0, 2, 1, 3)) < 4e-3
    vt_nat = torch.zeros_like(vt)
    vt_nat[..., ops.vt_key_order(tp, dev)] = vt                 # undo the key permutation of the V^T layout
    assert rel_l2(vt_nat[:, :, :, :T].float(), ref[:, :, 2].permute(0, 2, 3, 1)) < 4e-3
    assert float(q[:, :, T:].abs().max()) == 0 and float(vt_nat[:, :, :, T:].abs().max()) == 0


@pytest.mark.parametrize("B,T,H,K", [(2, 768, 16, 1152), (3, 96, 8, 128)])
def test_gemm_heads_split_head_size_72_padded(ops, B, T, H, K):
    """DiT-XL/2 head split: 72-wide heads written into zero-initialised 128-wide rows (q, k) / 128 rows (V^T).  r4: the staged,
    head-aware epilogue serves it (8-feature chunks lie inside one head) instead of the scattered direct stores; every tile
    configuration must agree with fp32 torch and leave the padding untouched."""
    dev, Dh, Dp = 'cuda', 72, 128
    g = torch.Generator().manual_seed(B * 10 + T)
    x = torch.randn(B * T, K, generator=g).to(dev)
    w = (torch.randn(3 * H * Dh, K, generator=g) * 0.1).to(dev)
    b = torch.randn(3 * H * Dh, generator=g).to(dev)
    xb, wb = _bf(x), _bf(w)
    ref = (xb.float() @ wb.float().t() + b).reshape(B, T, 3, H, Dh)
    q = torch.zeros(B, H, T, Dp, device=dev, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, Dp, T, device=dev, dtype=torch.bfloat16)
    ops.gemm(xb, wb, b, ops.EPI_HEADS, q, k, vt, M=B * T, tokens=T, tok_pad=T, heads=H, head_dim=Dh, transpose_mask=0b100, head_dim_pad=Dp)
    assert rel_l2(q[..., :Dh].float(), ref[:, :, 0].permute(0, 2, 1, 3)) < 4e-3
    assert rel_l2(k[..., :Dh].float(), ref[:, :, 1].permute(0, 2, 1, 3)) < 4e-3
    vt_nat = torch.zeros_like(vt)
    vt_nat[..., ops.vt_key_order(T, dev)] = vt
    assert rel_l2(vt_nat[:, :, :Dh].float(), ref[:, :, 2].permute(0, 2, 3, 1)) < 4e-3
    assert float(q[..., Dh:].abs().max()) == 0 and float(k[..., Dh:].abs().max()) == 0 and float(vt[:, :, Dh:].abs().max()) == 0


@pytest.mark.parametrize("B,T,H,K", [(4, 768, 16, 1024), (2, 1024, 4, 256), (3, 800, 2, 128)])
def test_gemm_heads_split_with_fused_qk_norm(ops, B, T, H, K):
    """HEADS epilogue with head_norm0 / head_norm1: q, k = RMSNorm_64(x W^T + b) * w per (token, head), one rounding to bf16;
    V^T untouched.  Shapes take the 384x192, 256x192 and 128x384 head-aligned tiles; small / unaligned problems must refuse."""
    dev = 'cuda'
    Dh = 64
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.randn(B * T, K, generator=g).to(dev)
    w = (torch.randn(3 * H * Dh, K, generator=g) * 0.1).to(dev)
    b = torch.randn(3 * H * Dh, generator=g).to(dev)
    nq = (1 + 0.3 * torch.randn(Dh, generator=g)).to(dev)
    nk = (1 + 0.3 * torch.randn(Dh, generator=g)).to(dev)
    xb, wb = _bf(x), _bf(w)
    import os
    forced = os.environ.get('LN3D_GEMM_TILE')
    small = B * T < 1536                                  # below the ring kernels' size: the 128x128 kernel has no fused qk_norm
    assert ops.heads_norm_fusable(B * T, 3 * H * Dh, T, Dh) == (forced in (None, 'x8', 'x9', 'x12', 'x14') and not (small and forced is None))
    if forced not in (None, 'x8', 'x9', 'x12', 'x14') or (small and forced is None):          # tiles without the head-aligned epilogue refuse
        with pytest.raises(RuntimeError):
            ops.gemm(xb, wb, b, ops.EPI_HEADS, torch.zeros(B, H, T, Dh, device=dev, dtype=torch.bfloat16), None, None, M=B * T, tokens=T,
                     tok_pad=T, heads=H, head_dim=Dh, head_norm0=nq)
        return
    ref = (xb.float() @ wb.float().t() + b).reshape(B, T, 3, H, Dh)
    rms = lambda t, wt: t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-5) * wt
    q = torch.zeros(B, H, T, Dh, device=dev, dtype=torch.bfloat16)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, Dh, T, device=dev, dtype=torch.bfloat16)
    ops.gemm(xb, wb, b, ops.EPI_HEADS, q, k, vt, M=B * T, tokens=T, tok_pad=T, heads=H, head_dim=Dh, transpose_mask=0b100,
             head_norm0=nq, head_norm1=nk, head_norm_eps=1e-5)
    assert rel_l2(q.float(), rms(ref[:, :, 0], nq).permute(0, 2, 1, 3)) < 4e-3
    assert rel_l2(k.float(), rms(ref[:, :, 1], nk).permute(0, 2, 1, 3)) < 4e-3
    vt_nat = torch.zeros_like(vt)
    vt_nat[..., ops.vt_key_order(T, dev)] = vt
    assert rel_l2(vt_nat.float(), ref[:, :, 2].permute(0, 2, 3, 1)) < 4e-3
    # agrees with the two-kernel route (GEMM, then ln3d_rmsnorm_heads_bf16 on the rounded q) to bf16 rounding
    q2 = torch.zeros_like(q)
    k2 = torch.zeros_like(q)
    ops.gemm(xb, wb, b, ops.EPI_HEADS, q2, k2, vt, M=B * T, tokens=T, tok_pad=T, heads=H, head_dim=Dh, transpose_mask=0b100)
    ops.rmsnorm_heads(q2, nq, B * H * T, Dh)
    assert rel_l2(q.float(), q2.float()) < 6e-3
    # the small-problem kernel has no such epilogue: refused, not silently un-normalised
    if forced is None:
      with pytest.raises(RuntimeError):
        ops.gemm(xb[:512], wb, b, ops.EPI_HEADS, q, k, vt, M=512, tokens=256, tok_pad=T, heads=H, head_dim=Dh, transpose_mask=0b100,
                 head_norm0=nq)


def test_gemm_gelu_erf_epilogue_tail(hip_lib, auto_tile):
    """the GELU-erf epilogue is a polynomial erf.  Against torch's exact erf-GELU on inputs that cover the negative tail
    ([-6, 6], every fc1 / ViT / CLIP / DINO MLP goes through it): absolute error <= 1.5e-4 + the bf16 rounding of the output, and the
    result has the sign of its argument (the r4 form returned +2e-6 below x = -3.9987)."""
    from ln3diff_amd import ops
    dev = 'cuda'
    M, K, N = 2048, 64, 256
    x = torch.zeros(M, K, device=dev)
    vals = torch.linspace(-6, 6, M * N // 1, device=dev)[:M * 4].reshape(M, 4)      # 4 probe values per row, routed by one-hot weights
    grid = torch.linspace(-6.0, 6.0, M * 4, device=dev).reshape(M, 4)
    x[:, :4] = grid
    w = torch.zeros(N, K, device=dev)
    w[torch.arange(N), torch.arange(N) % 4] = 1.0                                  # feature n reads probe n % 4
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(xb, wb, None, ops.EPI_GELU_ERF, out)
    arg = xb.float()[:, :4].repeat(1, N // 4).reshape(M, N // 4, 4).reshape(M, N)   # feature n's argument = probe n % 4
    arg = xb.float()[:, torch.arange(N, device=dev) % 4]
    ref = torch.nn.functional.gelu(arg.double()).float()
    got = out.float()
    err = (got - ref).abs()
    bound = 1.5e-4 + ref.abs() * 2.0 ** -8
    assert bool((err <= bound).all()), float((err - bound).max())
    neg, pos = arg < 0, arg > 0
    assert bool((got[neg] <= 0).all()) and bool((got[pos] >= 0).all())
    tail = arg <= -4.0
    assert tail.any() and float(got[tail].abs().max()) <= 3e-5


def _attn_ref(q, k, v, scale):
    s = (q.float() @ k.float().transpose(-1, -2)) * scale
    return torch.softmax(s, -1) @ v.float()


@pytest.mark.parametrize("B,H,Nq,Nk,Dh", [(2, 2, 768, 768, 64), (1, 3, 768, 77, 64), (2, 2, 256, 256, 64),
                                           (1, 2, 768, 1024, 64), (3, 1, 256, 256, 128), (1, 1, 100, 130, 64)])
def test_attention(ops, B, H, Nq, Nk, Dh):
    dev = 'cuda'
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    nqp, nkp = (Nq + 63) // 64 * 64, (Nk + 63) // 64 * 64
    q = torch.zeros(B, H, nqp, Dh)
    k = torch.zeros(B, H, nkp, Dh)
    v = torch.zeros(B, H, nkp, Dh)
    q[:, :, :Nq] = torch.randn(B, H, Nq, Dh, generator=g) * 1.5
    k[:, :, :Nk] = torch.randn(B, H, Nk, Dh, generator=g) * 1.5
    v[:, :, :Nk] = torch.randn(B, H, Nk, Dh, generator=g) + torch.arange(Dh) / Dh     # asymmetric in d
    # force the online-softmax rescale: a late key block with a huge score for some query rows
    if Nk > 128:
        k[:, :, Nk - 5] = q[:, :, 3] * 4.0
    qb, kb, vb = (_bf(t).to(dev) for t in (q, k, v))
    vt = vb.transpose(-1, -2)[..., ops.vt_key_order(nkp, dev)].contiguous()      # key-permuted V^T layout (ABI)
    out = torch.empty(B, Nq, H * Dh, device=dev, dtype=torch.bfloat16)
    ops.attention(qb, kb, vt, out, B, H, Nq, nqp, Nk, nkp, Dh)
    ref = _attn_ref(qb[:, :, :Nq], kb[:, :, :Nk], vb[:, :, :Nk], Dh ** -0.5)       # [B,H,Nq,Dh]
    ref = ref.permute(0, 2, 1, 3).reshape(B, Nq, H * Dh)
    e = rel_l2(out.float(), ref)
    assert e < 1e-2, e


@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 16, 768, 768), (1, 3, 300, 1024), (16, 16, 768, 768)])
def test_attention_head_size_72_in_padded_rows(hip_lib, B, H, Nq, Nk):
    """r4 (ABI 8): DiT-XL/2 heads (72) stored zero-padded to 128.  The kernel contracts over 80 dims, produces 96 output rows and
    writes COMPACT [B, Nq, H * 72] heads.  Against fp32 torch on the same bf16 operands, against the padded-128 run of the same
    kernel family (same values, other layout), spiked keys force the rescale branch; the largest case prints both timings."""
    from ln3diff_amd import ops
    dev, Dh, Dp = 'cuda', 72, 128
    g = torch.Generator().manual_seed(Nq + Nk + B)
    nqp, nkp = (Nq + 63) // 64 * 64, (Nk + 63) // 64 * 64
    q = torch.zeros(B, H, nqp, Dp); k = torch.zeros(B, H, nkp, Dp); v = torch.zeros(B, H, nkp, Dp)
    q[:, :, :Nq, :Dh] = torch.randn(B, H, Nq, Dh, generator=g) * 1.5
    k[:, :, :Nk, :Dh] = torch.randn(B, H, Nk, Dh, generator=g) * 1.5
    v[:, :, :Nk, :Dh] = torch.randn(B, H, Nk, Dh, generator=g) + torch.arange(Dh) / Dh
    k[:, :, Nk - 5] = q[:, :, 3] * 4.0
    qb, kb, vb = (_bf(t).to(dev) for t in (q, k, v))
    vt = vb.transpose(-1, -2)[..., ops.vt_key_order(nkp, dev)].contiguous()
    out = torch.full((B, Nq, H * Dh), float('nan'), device=dev, dtype=torch.bfloat16)
    ops.attention(qb, kb, vt, out, B, H, Nq, nqp, Nk, nkp, Dp, scale=Dh ** -0.5, dh_true=Dh)
    ref = _attn_ref(qb[:, :, :Nq, :Dh], kb[:, :, :Nk, :Dh], vb[:, :, :Nk, :Dh], Dh ** -0.5).permute(0, 2, 1, 3).reshape(B, Nq, H * Dh)
    assert torch.isfinite(out.float()).all()
    e = rel_l2(out.float(), ref)
    assert e < 1e-2, e
    full = torch.empty(B, Nq, H * Dp, device=dev, dtype=torch.bfloat16)
    ops.attention(qb, kb, vt, full, B, H, Nq, nqp, Nk, nkp, Dp, scale=Dh ** -0.5)
    full = full.reshape(B, Nq, H, Dp)
    assert float(full[..., Dh:].abs().max()) == 0.0
    assert rel_l2(out.float(), full[..., :Dh].reshape(B, Nq, H * Dh).float()) < 2e-3
    out2 = torch.empty_like(out)
    ops.attention(qb, kb, vt, out2, B, H, Nq, nqp, Nk, nkp, Dp, scale=Dh ** -0.5, dh_true=Dh)
    assert torch.equal(out, out2)                                        # run-to-run determinism
    if B * H >= 256:
        def t(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20 * 1e3
        t72 = t(lambda: ops.attention(qb, kb, vt, out, B, H, Nq, nqp, Nk, nkp, Dp, scale=Dh ** -0.5, dh_true=Dh))
        t128 = t(lambda: ops.attention(qb, kb, vt, full, B, H, Nq, nqp, Nk, nkp, Dp, scale=Dh ** -0.5))
        print(f"XL/2 self-attention {B}x{H} heads {Nq}x{Nk}: Dh_true 72 {t72:.1f} us, padded 128 {t128:.1f} us")
        assert t72 < t128


@pytest.mark.parametrize("B,H,Nq,Nk,Dh", [(2, 16, 768, 768, 72), (1, 3, 300, 1000, 72), (2, 5, 768, 1024, 80), (1, 2, 333, 77, 80), (16, 16, 768, 768, 72)])
def test_attention_heads_stored_80_wide(hip_lib, B, H, Nq, Nk, Dh):
    """r6: heads of 65 - 80 dims are stored 80 wide (attn_kernel<80, 2, DT>: 176-byte K pitch in LDS, 11 + 10 DMA instructions per stage dealt
    3 / 3 / 3 / 3 / 3 / 2 / 2 / 2 to the waves with per-wave counted waits).  DiT-XL/2's 72 (compact [B, Nq, H * 72] output) and true 80-wide
    heads, ragged query / key counts, a spiked key for the rescale branch: against fp32 torch on the same bf16 operands, against the 128-wide
    storage of the same values (72), bitwise repeat; the XL/2-size case prints both timings and must be faster than the 128-wide one."""
    from ln3diff_amd import ops
    dev, Dp = 'cuda', 80
    g = torch.Generator().manual_seed(Nq + Nk + B + Dh)
    nqp, nkp = (Nq + 63) // 64 * 64, (Nk + 63) // 64 * 64

    def operands(width):
        gg = torch.Generator().manual_seed(Nq + Nk + B + Dh)
        q = torch.zeros(B, H, nqp, width); k = torch.zeros(B, H, nkp, width); v = torch.zeros(B, H, nkp, width)
        q[:, :, :Nq, :Dh] = torch.randn(B, H, Nq, Dh, generator=gg) * 1.5
        k[:, :, :Nk, :Dh] = torch.randn(B, H, Nk, Dh, generator=gg) * 1.5
        v[:, :, :Nk, :Dh] = torch.randn(B, H, Nk, Dh, generator=gg) + torch.arange(Dh) / Dh
        k[:, :, Nk - 5] = q[:, :, 3] * 4.0
        qb, kb, vb = (_bf(t).to(dev) for t in (q, k, v))
        vt = vb.transpose(-1, -2)[..., ops.vt_key_order(nkp, dev)].contiguous()
        return qb, kb, vb, vt

    qb, kb, vb, vt = operands(Dp)
    dt = Dh if Dh != Dp else 0
    out = torch.full((B, Nq, H * Dh), float('nan'), device=dev, dtype=torch.bfloat16)
    run80 = lambda o: ops.attention(qb, kb, vt, o, B, H, Nq, nqp, Nk, nkp, Dp, scale=Dh ** -0.5, dh_true=dt)
    run80(out)
    ref = _attn_ref(qb[:, :, :Nq, :Dh], kb[:, :, :Nk, :Dh], vb[:, :, :Nk, :Dh], Dh ** -0.5).permute(0, 2, 1, 3).reshape(B, Nq, H * Dh)
    assert torch.isfinite(out.float()).all()
    e = rel_l2(out.float(), ref)
    assert e < 1e-2, e
    out2 = torch.empty_like(out)
    run80(out2)
    assert torch.equal(out, out2)                                        # run-to-run determinism
    if Dh == 72:                                                         # the same values in 128-wide rows: the r4 / r5 storage
        q1, k1, _, vt1 = operands(128)
        o128 = torch.empty_like(out)
        run128 = lambda: ops.attention(q1, k1, vt1, o128, B, H, Nq, nqp, Nk, nkp, 128, scale=Dh ** -0.5, dh_true=Dh)
        run128()
        assert rel_l2(out.float(), o128.float()) < 2e-3
        if B * H >= 256:
            def t(fn):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 20 * 1e3
            t80, t128 = t(lambda: run80(out)), t(run128)
            print(f"XL/2 self-attention {B}x{H} heads {Nq}x{Nk}: 80-wide storage {t80:.1f} us, 128-wide {t128:.1f} us")
            assert t80 < t128


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 1, 256, 256), (1, 3, 512, 512), (5, 8, 768, 768), (19, 16, 768, 1024), (1, 2, 1280, 1280),
                                       (2, 1, 2048, 2048), (1, 12, 3072, 3072), (3, 2, 200, 512), (2, 3, 768, 2304)])
def test_attention_streaming_kernel_shapes(ops, B, H, Nq, Nk):
    """attn_stream_kernel (Dh 64, Nk a multiple of 256): ring wrap at 1, 2, 3 .. 12 query / key blocks, head counts that do and do
    not fill the chip, fewer queries than keys (the I23D appended-token layout), a spiked late key (deferred re-base), and the result
    repeated bit for bit."""
    dev = 'cuda'
    g = torch.Generator().manual_seed(Nq * 13 + Nk + B)
    nqp = (Nq + 63) // 64 * 64
    q = torch.zeros(B, H, nqp, 64)
    q[:, :, :Nq] = torch.randn(B, H, Nq, 64, generator=g) * 1.5
    k = torch.randn(B, H, Nk, 64, generator=g) * 1.5
    v = torch.randn(B, H, Nk, 64, generator=g) + torch.arange(64) / 64
    k[:, :, Nk - 37] = q[:, :, 5] * 4.0
    k[:, :, 11] = q[:, :, Nq - 1] * 3.0
    qb, kb, vb = (_bf(t).to(dev) for t in (q, k, v))
    vt = vb.transpose(-1, -2)[..., ops.vt_key_order(Nk, dev)].contiguous()
    outs = []
    for _ in range(2):
        out = torch.empty(B, Nq, H * 64, device=dev, dtype=torch.bfloat16)
        ops.attention(qb, kb, vt, out, B, H, Nq, nqp, Nk, Nk, 64)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    ref = _attn_ref(qb[:, :, :Nq], kb, vb, 64 ** -0.5).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    e = rel_l2(outs[0].float(), ref)
    assert e < 1e-2, e


@pytest.mark.parametrize("B,H,Nq,Nk,spike", [(16, 16, 768, 768, 4.0), (16, 16, 768, 768, 40.0), (32, 8, 512, 512, 40.0), (19, 14, 768, 512, 3.0),
                                             (16, 16, 256, 768, 40.0)])
def test_attention_k_resident_kernel(ops, B, H, Nq, Nk, spike):
    """attn_kres_kernel (Dh 64, Nk 512 / 768, Nq a multiple of 256, a head for every CU = the benchmarked DiT-L/2 geometry,
    network batch 16 x 16 heads): K resident in LDS, row sums on the matrix pipe, fixed reference per query block.  spike 40:
    one key per head scores ~2^340 above the reference of its row, which overflows the fast path, is detected through the row
    sum and recomputed exactly in-kernel (only every third head is spiked, so both paths run in one launch); an early-tile spike
    makes every other score of a row underflow.  Compared with fp32 softmax attention on the same bf16 operands, bit-repeatable."""
    dev = 'cuda'
    g = torch.Generator().manual_seed(Nq * 13 + Nk + B)
    q = torch.randn(B, H, Nq, 64, generator=g) * 1.5
    k = torch.randn(B, H, Nk, 64, generator=g) * 1.5
    v = torch.randn(B, H, Nk, 64, generator=g) + torch.arange(64) / 64
    k[:, ::3, Nk - 37] = q[:, ::3, 5] * spike                  # late key: far above the first tile's maximum
    k[:, 1::3, Nk // 2 + 3] = q[:, 1::3, Nq - 2] * 4.0
    k[:, :, 11] = q[:, :, Nq - 1] * spike                        # first tile: the reference itself is the spike
    qb, kb, vb = (_bf(t).to(dev) for t in (q, k, v))
    vt = vb.transpose(-1, -2)[..., ops.vt_key_order(Nk, dev)].contiguous()
    outs = []
    for _ in range(2):
        out = torch.empty(B, Nq, H * 64, device=dev, dtype=torch.bfloat16)
        ops.attention(qb, kb, vt, out, B, H, Nq, Nq, Nk, Nk, 64)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert torch.isfinite(outs[0].float()).all()
    ref = _attn_ref(qb, kb, vb, 64 ** -0.5).permute(0, 2, 1, 3).reshape(B, Nq, H * 64)
    e = rel_l2(outs[0].float(), ref)
    assert e < 1e-2, e
    # the spiked rows themselves (softmax ~ one-hot): row 5 of every third head must equal V of the spiked key
    row = outs[0].float().view(B, Nq, H, 64)[:, 5, ::3]
    tgt = vb.float()[:, ::3, Nk - 37]
    assert rel_l2(row, tgt) < 1e-2


@pytest.mark.parametrize("D,kind", [(128, 0), (768, 0), (1024, 0), (1152, 0), (1024, 1)])
def test_norm_modulate(ops, D, kind):
    dev = 'cuda'
    B, N = 3, 40
    g = torch.Generator().manual_seed(D)
    x = (torch.randn(B * N, D, generator=g) * 2 + 0.5).to(dev)
    mod = torch.randn(B, 6 * D, generator=g).to(dev)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    y = torch.empty(B * N, D, device=dev, dtype=torch.bfloat16)
    sh, sc = mod[:, D:], mod[:, 2 * D:]
    if kind == 0:
        ops.norm_modulate(x, y, B * N, D, kind=0, eps=1e-6, shift=sh, scale=sc, mod_rows=N, mod_ld=6 * D)
        n = torch.nn.functional.layer_norm(x, (D,), eps=1e-6)
    else:
        ops.norm_modulate(x, y, B * N, D, kind=1, eps=1e-5, weight=w, shift=sh, scale=sc, mod_rows=N, mod_ld=6 * D)
        n = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * w
    ref = n * (1 + sc[:, :D].repeat_interleave(N, 0)) + sh[:, :D].repeat_interleave(N, 0)
    assert rel_l2(y.float(), ref) < 4e-3
    # row remap (append room): rows_in N -> rows_out N+8
    y2 = torch.zeros(B * (N + 8), D, device=dev, dtype=torch.bfloat16)
    ops.norm_modulate(x, y2, B * N, D, kind=0, eps=1e-6, rows_in=N, rows_out=N + 8)
    ref2 = torch.nn.functional.layer_norm(x, (D,), eps=1e-6).reshape(B, N, D)
    assert rel_l2(y2.reshape(B, N + 8, D)[:, :N].float(), ref2) < 4e-3
    assert float(y2.reshape(B, N + 8, D)[:, N:].abs().max()) == 0


def test_rmsnorm_heads(ops):
    dev = 'cuda'
    x = torch.randn(1000, 64, device=dev).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(64, device=dev))
    ref = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-5) * w
    y = x.clone()
    ops.rmsnorm_heads(y, w, 1000, 64)
    assert rel_l2(y.float(), ref) < 4e-3


def test_timestep_patch_final(ops):
    from oracle import dit as odit
    dev = 'cuda'
    t = torch.tensor([0.0, 1.0, 37.0, 999.0, 0.25])
    out = torch.empty(5, 256, device=dev, dtype=torch.bfloat16)
    ops.timestep_embedding(t.to(dev), out, 5, 256)
    assert rel_l2(out.float(), odit.timestep_embedding(t)) < 4e-3
    # patch embed vs oracle
    D, C, S, p = 128, 4, 32, 2
    g = torch.Generator().manual_seed(0)
    sd = {'x_embedder.proj.weight': torch.randn(D, C, p, p, generator=g), 'x_embedder.proj.bias': torch.randn(D, generator=g)}
    pos = odit.trilatent_pos_embed(D)
    x = torch.randn(2, 12, S, S, generator=g)
    ref = odit.patchify_embed(sd, torch.cat([x, x]) * torch.tensor([1., 1., .5, .5]).view(4, 1, 1, 1)) + pos
    tok = torch.empty(4 * 768, D, device=dev)
    ops.patch_embed(x.to(dev), torch.tensor([1., 1., .5, .5], device=dev), sd['x_embedder.proj.weight'].reshape(D, -1).to(dev),
                    sd['x_embedder.proj.bias'].to(dev), pos[0].to(dev), tok, 2, 4, C, S, p, D)
    assert rel_l2(tok.reshape(4, 768, D), ref) < 1e-5
    # final layer vs oracle math
    h = torch.randn(2, 768, D, generator=g)
    mod = torch.randn(2, 2 * D, generator=g)
    wf, bf_ = torch.randn(16, D, generator=g) * 0.1, torch.randn(16, generator=g)
    yr = torch.nn.functional.layer_norm(h, (D,), eps=1e-6) * (1 + mod[:, None, D:]) + mod[:, None, :D]
    yr = odit.unpatchify_trilatent(torch.nn.functional.linear(yr, wf, bf_), 2, 2, 4)
    o = torch.empty(2, 12, 32, 32, device=dev)
    md = mod.to(dev)
    ops.final_layer(h.to(dev).reshape(-1, D), md[:, :D], md[:, D:], 2 * D, None, None, wf.to(dev), bf_.to(dev), o, 2, 4, 32, 2, D)
    assert rel_l2(o, yr) < 1e-5
    # few tokens (3 x 16): the kernel without the LDS weight image; same arithmetic
    h8 = torch.randn(2, 48, D, generator=g)
    y8 = torch.nn.functional.layer_norm(h8, (D,), eps=1e-6) * (1 + mod[:, None, D:]) + mod[:, None, :D]
    y8 = odit.unpatchify_trilatent(torch.nn.functional.linear(y8, wf, bf_), 2, 2, 4)            # [2, 12, 8, 8]
    o8 = torch.empty(2, 12, 8, 8, device=dev)
    ops.final_layer(h8.to(dev).reshape(-1, D), md[:, :D], md[:, D:], 2 * D, None, None, wf.to(dev), bf_.to(dev), o8, 2, 4, 8, 2, D)
    assert rel_l2(o8, y8) < 1e-5


def test_sampler_steps(ops):
    dev = 'cuda'
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 12, 32, 32, generator=g)
    eps2 = torch.randn(4, 12, 32, 32, generator=g)
    sig, nxt, s = 3.7, 3.1, 6.5
    du, dc = (x - sig * eps2[:2]), (x - sig * eps2[2:])
    d = du + s * (dc - du)
    ref = x + (x - d) / sig * (nxt - sig)
    xd = x.to(dev).clone()
    ops.edm_euler_step(xd, eps2.to(dev), sig, nxt, s)
    assert rel_l2(xd, ref) < 1e-5
    v2 = torch.randn(4, 12, 32, 32, generator=g)
    x2 = torch.cat([x, x]).to(dev)
    ops.flow_euler_step(x2, v2.to(dev), 0.02, 4.0)
    v = v2[2:] + 4.0 * (v2[:2] - v2[2:])
    assert rel_l2(x2[:2], x + 0.02 * v) < 1e-6 and torch.equal(x2[:2], x2[2:])


@pytest.mark.parametrize("Lc,H,Bn", [(77, 16, 2), (96, 12, 2), (33, 4, 2), (77, 16, 16), (77, 16, 8)])
def test_gemm_cross_attention_epilogue(ops, Lc, H, Bn):
    """LN3D_EPI_CROSS_ATTN (query projection + attention over a short cached context in one kernel) vs a torch fp32
    reference on the same bf16-rounded operands, and vs the unfused HIP path (HEADS GEMM + ln3d_attention_bf16).  Bn = 16 is the
    benchmarked network batch (256x192 tiles, 8 waves); Bn = 8 - the conditional half of it - and Bn = 2 leave that tiling
    under-filled and take the 128x192 / 4-wave form (2 heads per tile)."""
    torch.manual_seed(0)
    dev = 'cuda'
    N, K = 768, 1024
    M, D = Bn * N, H * 64
    lpad = (Lc + 63) // 64 * 64
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(D, K, device=dev) * 0.03).to(torch.bfloat16)
    kc = torch.zeros(Bn, H, lpad, 64, device=dev, dtype=torch.bfloat16)
    vc = torch.zeros(Bn, H, lpad, 64, device=dev, dtype=torch.bfloat16)
    kc[:, :, :Lc] = torch.randn(Bn, H, Lc, 64, device=dev).to(torch.bfloat16)
    vc[:, :, :Lc] = torch.randn(Bn, H, Lc, 64, device=dev).to(torch.bfloat16)
    vt = vc.transpose(-1, -2).contiguous()[..., ops.vt_key_order(lpad, dev)].contiguous()
    kp = kc[..., ops.vt_key_order(64, dev)].contiguous()
    out = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    ops.gemm(x, w, None, ops.EPI_CROSS_ATTN, out, kp, vt, M=M, tokens=N, heads=H, head_dim=64, ctx_keys=Lc, ctx_pad=lpad,
             ctx_scale=0.125)
    # torch reference
    q = (x.float() @ w.float().t()).to(torch.bfloat16).float().view(Bn, N, H, 64).transpose(1, 2)
    a = torch.softmax(q @ kc[:, :, :Lc].float().transpose(-1, -2) * 0.125, -1) @ vc[:, :, :Lc].float()
    ref = a.transpose(1, 2).reshape(M, D)
    e = float((out.float() - ref).norm() / ref.norm())
    # unfused HIP path
    qh = torch.zeros(Bn, H, N, 64, device=dev, dtype=torch.bfloat16)
    ops.gemm(x, w, None, ops.EPI_HEADS, qh, M=M, tokens=N, tok_pad=N, heads=H, head_dim=64)
    o2 = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    ops.attention(qh, kc, vt, o2, Bn, H, N, N, Lc, lpad, 64)
    e2 = float((out.float() - o2.float()).norm() / o2.float().norm())
    print('cross-attn epilogue', Lc, H, 'vs torch', e, 'vs unfused', e2)
    assert e < 1e-2 and e2 < 5e-3, (e, e2)
