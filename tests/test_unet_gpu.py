"""GPU parity of the U-Net denoiser (ShapeNet / FFHQ entry point) on the HIP kernels: forward against the reference's
UNetModel goldens (tiny spatial-transformer config, AttentionBlock + roll_out config, the released ShapeNet size with 827 M
parameters), the DDIM loop with v-prediction + LSGM mixed prediction against the reference's own loop, and the glue kernels of
csrc/unet_ops.hip against fp32 torch."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, manifest, rel_l2

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
pytestmark = pytest.mark.gpu


def _model(tag):
    from test_unet_cpu import _product
    from unet_configs import CONFIGS, synth_unet_sd
    cfg = CONFIGS[tag]
    m = _product(cfg)
    m.load_state_dict(synth_unet_sd(manifest(golden('unet_' + tag)), 0), strict=True)
    return m.cuda(), cfg


@pytest.mark.parametrize("tag", ['tiny_st', 'tiny_attn', 'shapenet'])
def test_unet_forward_vs_reference_golden(hip_lib, tag):
    from ln3diff_amd.synth import synth_input
    g = golden('unet_' + tag)
    m, cfg = _model(tag)
    B = g['y'].shape[0]
    C = cfg['in_channels'] * (3 if cfg['roll_out'] else 1)
    x = synth_input('x', (B, C, cfg['image_size'], cfg['image_size']), 3).cuda()
    ctx = synth_input('c', (B, 77, cfg['context_dim']), 3).cuda() if cfg['use_spatial_transformer'] else None
    y = m(x, torch.from_numpy(g['t']).cuda(), context=ctx)
    y2 = m(x, torch.from_numpy(g['t']).cuda(), context={'crossattn': ctx} if ctx is not None else None)      # sgm conditioner dict (unet.py:762)
    assert y.shape == g['y'].shape and torch.equal(y, y2)
    e = rel_l2(y.cpu(), g['y'])
    print('unet', tag, e)
    assert e < 2e-2, e                     # bf16 GEMM operands against the reference's fp32 (the DiT forwards: 1e-3 .. 3e-3)


@pytest.mark.parametrize("spec", ['ddim25', 'ddim10'])
def test_unet_ddim_v_prediction_mixing_vs_reference_golden(hip_lib, spec):
    """SpacedDiffusion.ddim_sample_loop over the U-Net as the ShapeNet entry point runs it: ModelMeanType.V, mixing_normal=True,
    classifier-free guidance with the zero embedding; noise stream = the reference's (seeded randn per step)."""
    from ln3diff_amd.guided_diffusion import gaussian_diffusion as gd
    from ln3diff_amd.guided_diffusion.respace import SpacedDiffusion, space_timesteps
    from ln3diff_amd.synth import synth_input
    g = golden('unet_ddim_tiny_' + spec)
    m, cfg = _model('tiny_st')
    B = 2
    z = synth_input('z', (B, 4, 16, 16), 41).cuda()
    c = synth_input('c', (B, 77, 768), 41).cuda()
    diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, spec), betas=gd.get_named_beta_schedule('linear', 1000),
                           model_mean_type=gd.ModelMeanType.V, model_var_type=gd.ModelVarType.FIXED_LARGE, rescale_timesteps=False)
    torch.manual_seed(int(g['noise_seed']))
    noises = [torch.randn(B, 4, 16, 16) for _ in range(diff.num_timesteps)]
    y = diff.ddim_sample_loop(m, (B, 4, 16, 16), cond={'c_crossattn': c}, noise=z.clone(), clip_denoised=False, device='cuda',
                              eta=float(g['eta']), unconditional_guidance_scale=float(g['scale']),
                              unconditional_conditioning=torch.zeros(1, 77, 768, device='cuda'), mixing_normal=True,
                              step_noise=lambda k: noises[k])
    e = rel_l2(y.cpu(), g['final'])
    print('unet ddim', spec, e)
    assert e < 2e-2, e


def test_unet_glue_kernels_vs_torch(hip_lib):
    from ln3diff_amd import ops
    import torch.nn.functional as F
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    # GroupNorm at C = 320 (10 channels per group), with the per-sample row / scale-shift modulation
    N, H, W, C = 2, 8, 8, 320
    x = torch.randn(N, H * W, C, generator=g).to(dev)
    w, b = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    row, sc, sh = (torch.randn(N, C, generator=g).to(dev) for _ in range(3))
    y = torch.empty(N * H * W, C, device=dev, dtype=torch.bfloat16)
    nchw = lambda t: t.reshape(N, H, W, C).permute(0, 3, 1, 2)
    ops.groupnorm_any(x, w, b, y, N, H * W, C, 32, 1e-5, True, add_row=row)
    ref = F.silu(F.group_norm(nchw(x + row[:, None]), 32, w, b, 1e-5))
    assert rel_l2(nchw(y.float()), ref) < 5e-3
    ops.groupnorm_any(x, w, b, y, N, H * W, C, 32, 1e-6, False, mod_scale=sc, mod_shift=sh)
    ref = F.group_norm(nchw(x), 32, w, b, 1e-6) * (1 + sc[:, :, None, None]) + sh[:, :, None, None]
    assert rel_l2(nchw(y.float()), ref) < 5e-3
    # strided im2col == F.unfold of the padded input with stride 2, in (ky, kx, c) column order
    Cc = 16
    xi = torch.randn(N, H, W, Cc, generator=g).to(dev).to(torch.bfloat16)
    col = torch.empty(N * 16, 192, device=dev, dtype=torch.bfloat16)
    ops.im2col3x3_strided(xi, col, N, H, W, Cc, 2, 192)
    un = F.unfold(xi.float().permute(0, 3, 1, 2), 3, padding=1, stride=2)            # [N, C*9, 16], rows (c, ky, kx)
    un = un.reshape(N, Cc, 9, 16).permute(0, 3, 2, 1).reshape(N * 16, 9 * Cc)
    assert torch.equal(col[:, :144].float(), un) and float(col[:, 144:].abs().max()) == 0.0
    # GEGLU
    xg = torch.randn(37, 2 * 96, generator=g).to(dev)
    yg = torch.empty(37, 96, device=dev, dtype=torch.bfloat16)
    ops.geglu(xg, yg, 37, 96)
    assert rel_l2(yg.float(), xg[:, :96] * F.gelu(xg[:, 96:])) < 5e-3
    # attention at head size 160 over 77 keys, q / k / v as column slices of wider rows
    B, Hh, Nq, Nk, Dh = 2, 3, 16, 77, 160
    q = torch.randn(B * Nq, Hh * Dh, generator=g).to(dev).to(torch.bfloat16)
    kv = torch.randn(B * Nk, 2 * Hh * Dh, generator=g).to(dev).to(torch.bfloat16)
    o = torch.empty(B * Nq, Hh * Dh, device=dev, dtype=torch.bfloat16)
    ops.attention_small(q, kv, kv[:, Hh * Dh:], o, B, Hh, Nq, Nk, Dh, Hh * Dh, 2 * Hh * Dh, 2 * Hh * Dh, Dh ** -0.5)
    sp = lambda t, n: t.float().reshape(B, n, Hh, Dh).permute(0, 2, 1, 3)
    ref = (torch.softmax(sp(q, Nq) @ sp(kv[:, :Hh * Dh], Nk).transpose(-1, -2) * Dh ** -0.5, -1) @ sp(kv[:, Hh * Dh:], Nk))
    assert rel_l2(o.float().reshape(B, Nq, Hh, Dh).permute(0, 2, 1, 3), ref) < 5e-3
    # layouts and the mixed prediction
    xn = torch.randn(2, 12, 5, 7, generator=g).to(dev)
    cl = torch.empty(2 * 35, 16, device=dev, dtype=torch.bfloat16)
    ops.nchw_to_cl_bf16(xn, cl, 2, 12, 35, 16)
    assert torch.equal(cl[:, :12].float().reshape(2, 35, 12).permute(0, 2, 1).reshape(2, 12, 5, 7), xn.to(torch.bfloat16).float())
    assert float(cl[:, 12:].abs().max()) == 0.0
    back = torch.empty(2, 12, 5, 7, device=dev)
    ops.cl_to_nchw_f32(xn.permute(0, 2, 3, 1).reshape(70, 12).contiguous(), back, 2, 12, 35)
    assert torch.equal(back, xn)
    eps, logit = torch.randn(2, 12, 5, 7, generator=g).to(dev), torch.linspace(-2, 2, 12).to(dev)
    want = (1 - torch.sigmoid(logit))[None, :, None, None] * (0.7 * xn) + torch.sigmoid(logit)[None, :, None, None] * eps
    ops.mix_prediction(eps, xn, logit, 0.7, 2, 12, 35)
    assert torch.allclose(eps, want, atol=1e-6)
