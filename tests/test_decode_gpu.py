"""GPU parity of the HIP tri-plane VAE decode (DiT2 backbone + conv decoder) against the reference golden.
bf16 GEMM operands, fp32 residual/accumulators -> rel-L2 <= 3e-2 on the planes (48 sequential bf16 GEMM layers
for DiT2-L/2 plus ~25 conv layers), tokens <= 2e-2."""
import pytest
import torch

from conftest import golden, load_synth, manifest, rel_l2

pytestmark = pytest.mark.gpu


def build_decoder(hidden, depth, heads):
    from ln3diff_amd.dit.dit_decoder import DiT2
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.vit.vit_triplane import (
        RodinSR_256_fusionv6_ConvQuant_liteSR_dinoInit3DAttn_SD_B_3L_C_withrollout_withSD_D_ditDecoder as Dec)
    vit = DiT2(input_size=16, patch_size=2, in_channels=hidden, hidden_size=hidden, depth=depth, num_heads=heads,
               num_classes=0, learn_sigma=False, mixed_prediction=False, context_dim=None, roll_out=True, plane_n=3)
    return Dec(vit_decoder=vit, triplane_decoder=Triplane(img_resolution=128), cls_token=False, vae_p=2,
               ldm_z_channels=4, ldm_embed_dim=4)


@pytest.mark.parametrize("tag,cfg,B", [('tiny', (128, 2, 2), 2), ('dit2_l2', (1024, 24, 16), 1)])
def test_vae_decode_vs_reference_golden(hip_lib, tag, cfg, B):
    from ln3diff_amd.synth import synth_input
    g = golden('decode_' + tag)
    dec = build_decoder(*cfg)
    ref_keys = manifest(g, 'all_keys')
    mine = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    assert mine == ref_keys, set(mine) ^ set(ref_keys)
    load_synth(dec, 0)
    dec = dec.cuda()
    latent = synth_input('latent', (B, 12, 32, 32), 5).cuda()
    tok = dec.vit_decode_backbone({'latent_normalized_2Ddiffusion': latent}, 128)
    e_tok = rel_l2(tok[:, ::16, ::8].cpu(), g['tokens_sub'])
    ret = dec.vit_decode_postprocess(tok, {})
    planes = ret['latent_after_vit']
    e_pl = rel_l2(planes[:, :, ::8, ::8].cpu(), g['planes_sub'])
    print(tag, 'tokens', e_tok, 'planes', e_pl, 'std', float(planes.std()), float(g['planes_std']))
    assert e_tok < 2e-2, e_tok
    assert e_pl < 3e-2, e_pl
    # channel-last planes (what the ray-marcher consumes) == reference layout, bit for bit
    pcl = ret['planes_channel_last']
    assert torch.equal(pcl.permute(0, 1, 4, 2, 3).reshape(B, 96, 128, 128), planes)


@pytest.mark.parametrize("tag,cfg,B", [('tiny', (128, 2, 2), 2), ('dit2_l2', (1024, 24, 16), 1)])
def test_vae_decode_backbone_vs_bf16_operand_oracle(hip_lib, tag, cfg, B):
    """The 2e-2 / 3e-2 gates above are the distance between bf16-operand and fp32 arithmetic, not a property of the kernels.
    Against the oracle evaluated WITH bf16 GEMM / attention operands and fp32 accumulation (oracle.dit.operand_rounding: the
    arithmetic the MFMA kernels implement) the DiT2 backbone's tokens agree to <= 3e-3 (measured 1e-4 tiny, 1.4e-3 DiT2-L/2: the
    fp32 residual stream with small gated branches damps rounding-boundary flips).  The conv decoder behind it does NOT damp them
    - every GroupNorm -> conv re-rounds the whole activation, two bf16 implementations decorrelate to the full bf16 noise level
    within a few layers (planes 1.6e-2 vs the bf16-operand oracle, the same as vs fp32) - so it is gated stage by stage below."""
    from ln3diff_amd.synth import synth_input
    from oracle import decoder as odec, dit as odit
    dec = build_decoder(*cfg)
    sd, _ = load_synth(dec, 0)
    dec = dec.cuda()
    latent = synth_input('latent', (B, 12, 32, 32), 5)
    tok = dec.vit_decode_backbone({'latent_normalized_2Ddiffusion': latent.cuda()}, 128)
    with odit.operand_rounding(torch.bfloat16):
        tok_or = odec.vae_decode(sd, latent, cfg[2], return_tokens=True)
    e_tok = rel_l2(tok.cpu().reshape(tok_or.shape), tok_or)
    print(tag, 'tokens vs bf16-operand oracle', e_tok)
    assert e_tok < 3e-3, e_tok


def test_conv_decoder_stages_vs_bf16_operand_oracle(hip_lib):
    """Every stage of the conv decoder (ResnetBlock with / without nin_shortcut, the attention block, nearest-2x upsample + conv,
    norm_out + conv_out; ldm/modules/diffusionmodules/model.py:94-153,209-275,54-70) on its own, HIP vs the bf16-operand oracle on
    the SAME input: <= 5e-4 per stage (measured 2e-7 ... 1.2e-4) (the end-to-end planes gate stays at 3e-2, see above)."""
    from ln3diff_amd import ops
    from oracle import decoder as odec, dit as odit
    dec = build_decoder(128, 2, 2)
    sd, _ = load_synth(dec, 0)
    dec = dec.cuda()
    dec._ensure_packed(torch.device('cuda'))
    P, pre = dec._packed, 'superresolution.conv_sr.'
    cl = lambda x: x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()                   # NCHW -> [N*H*W, C]
    nchw = lambda y, N, H, W: y.reshape(N, H, W, -1).permute(0, 3, 1, 2)
    g = torch.Generator().manual_seed(11)
    N = 3
    worst = 0.0
    stages = [('mid.block_1.', P['mid1'], 16), ('mid.block_2.', P['mid2'], 16), ('up.3.block.0.', P['up'][3]['blocks'][0], 16),
              ('up.2.block.0.', P['up'][2]['blocks'][0], 32), ('up.0.block.0.', P['up'][0]['blocks'][0], 32)]
    for name, q, H in stages:
        x = torch.randn(N, q['c1']['cin'], H, H, generator=g)
        y = dec._resblock(cl(x).cuda(), q, N, H, H).clone()
        with odit.operand_rounding(torch.bfloat16):
            y_or = odec.resnet_block(sd, pre + name, x)
        e = rel_l2(nchw(y.cpu(), N, H, H), y_or)
        print('resblock', name, tuple(y_or.shape), e)
        worst = max(worst, e)
    x = torch.randn(N, 128, 16, 16, generator=g)
    y = dec._attn(cl(x).cuda(), P['attn'], N, 16, 16).clone()
    with odit.operand_rounding(torch.bfloat16):
        y_or = odec.attn_block(sd, pre + 'mid.attn_1.', x)
    e = rel_l2(nchw(y.cpu(), N, 16, 16), y_or)
    print('attention block', e)
    worst = max(worst, e)
    u = P['up'][3]['upsample']
    x = torch.randn(N, u['cin'], 16, 16, generator=g)
    xb = torch.empty(N * 256, u['cin'], dtype=torch.bfloat16, device='cuda')
    ops.cast_bf16(cl(x).cuda(), xb)
    y = torch.empty(N * 1024, u['cout'], device='cuda')
    dec._conv3(xb, N, 16, 16, u, 2, y)
    with odit.operand_rounding(torch.bfloat16):
        y_or = odec._conv(torch.nn.functional.interpolate(x, scale_factor=2.0, mode='nearest'), sd, pre + 'up.3.upsample.conv.', 1)
    e = rel_l2(nchw(y.cpu(), N, 32, 32), y_or)
    print('upsample + conv', e)
    worst = max(worst, e)
    x = torch.randn(N, 32, 32, 32, generator=g)
    h = dec._gn(cl(x).cuda(), P['norm_out'], N, 1024, 32)
    y = torch.empty(N * 1024, 32, device='cuda')
    dec._conv3(h, N, 32, 32, P['conv_out'], 1, y)
    with odit.operand_rounding(torch.bfloat16):
        y_or = odec._conv(odec._swish(odec._gn(x, sd, pre + 'norm_out.')), sd, pre + 'conv_out.', 1)
    e = rel_l2(nchw(y.cpu(), N, 32, 32), y_or)
    print('norm_out + conv_out', e)
    worst = max(worst, e)
    assert worst < 5e-4, worst
