"""CPU oracle (TEST INFRASTRUCTURE ONLY) - fp32 torch restatement of the tri-plane VAE
decode (latent [B,12,32,32] -> planes [B,96,128,128]).  Never imported by the product.

Reference citations:
  PatchEmbedTriplane (grouped 2x2 conv + channel regroup)   vit/vit_triplane.py:58-110
  vit_decode_backbone                                       vit/vit_triplane.py:996-1011
  DiT2.forward / DiTBlock2 (per-token adaLN, in-plane/global alternation)
                                                            dit/dit_decoder.py:19-36,99-151
  decoder pos-embed init (grid (3p, p))                     vit/vit_triplane.py:333-343
  vit_decode_postprocess (token unflatten + conv Decoder)   vit/vit_triplane.py:1913-1976
  conv Decoder / ResnetBlock / Upsample / attn block        ldm/modules/diffusionmodules/model.py:625-745,94-153,54-70,209-275
State-dict names follow the reference's released decoder class
(`superresolution.ldm_upsample.*`, `vit_decoder.*`, `superresolution.conv_sr.*`).
"""
import torch
import torch.nn.functional as F

from .dit import OPERAND_ROUND, _r, layer_norm, linear, sdpa, self_attention, fused_mlp, sincos_pos_embed_2d


def decoder_pos_embed(D, p=16):
    pe = sincos_pos_embed_2d(D, (3 * p, p))
    return torch.from_numpy(pe).float().reshape(1, 3 * p * p, D)


def patch_embed_triplane(sd, latent, p='superresolution.ldm_upsample.'):
    w, b = sd[p + 'proj.weight'], sd[p + 'proj.bias']
    x = F.conv2d(latent, w, b, stride=w.shape[-1], groups=3)           # [B,3D,h,w]
    B = x.shape[0]
    x = x.reshape(B, x.shape[1] // 3, 3, x.shape[-2], x.shape[-1])      # literal regroup
    return x.flatten(2).transpose(1, 2)                                 # [B,3hw,D]


def dit2_block(sd, p, x, c, H):
    mod = linear(F.silu(c), sd[p + 'adaLN_modulation.1.weight'], sd[p + 'adaLN_modulation.1.bias'])
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=-1)
    x = x + g_a * self_attention(sd, p + 'attn.', layer_norm(x) * (1 + sc_a) + sh_a, H)
    return x + g_m * fused_mlp(sd, p + 'mlp.', layer_norm(x) * (1 + sc_m) + sh_m)


def dit2_forward(sd, c, num_heads, p='vit_decoder.', plane_n=3):
    B, L, D = c.shape
    depth = 1 + max(int(k[len(p):].split('.')[1]) for k in sd if k.startswith(p + 'blocks.'))
    x = sd[p + 'pos_embed'].repeat(B, 1, 1)
    cin = c.reshape(B * plane_n, L // plane_n, D)
    for i in range(depth):
        if i % 2 == 0:
            x = dit2_block(sd, f'{p}blocks.{i}.', x.reshape(B * plane_n, L // plane_n, D), cin, num_heads)
        else:
            x = dit2_block(sd, f'{p}blocks.{i}.', x.reshape(B, L, D), c, num_heads)
    return x.reshape(B, L, D)


def _gn(x, sd, p):
    return F.group_norm(x, 32, sd[p + 'weight'], sd[p + 'bias'], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(x, sd, p, pad):
    # under dit.operand_rounding(): bf16 activations / weights into an fp32-accumulating product, like the im2col + MFMA GEMM
    return F.conv2d(_r(x), _r(sd[p + 'weight']), sd[p + 'bias'], padding=pad)


def resnet_block(sd, p, x):
    h = _conv(_swish(_gn(x, sd, p + 'norm1.')), sd, p + 'conv1.', 1)
    h = _conv(_swish(_gn(h, sd, p + 'norm2.')), sd, p + 'conv2.', 1)
    if p + 'nin_shortcut.weight' in sd:
        x = _conv(x, sd, p + 'nin_shortcut.', 0)
    return x + h


def attn_block(sd, p, x):
    h = _gn(x, sd, p + 'norm.')
    q, k, v = (_conv(h, sd, p + n, 0) for n in ('q.', 'k.', 'v.'))
    B, C, Hh, Ww = q.shape
    f = lambda t: t.reshape(B, C, Hh * Ww).transpose(1, 2)              # [B,HW,C]
    if OPERAND_ROUND[0] is None:
        s = torch.softmax((f(q) @ f(k).transpose(1, 2)) * (C ** -0.5), dim=-1)
        o = (s @ f(v)).transpose(1, 2).reshape(B, C, Hh, Ww)
    else:                                                               # q / k / v stored in bf16, one 128-wide head (dit.sdpa's rounding)
        o = sdpa(_r(f(q))[:, None], _r(f(k))[:, None], _r(f(v))[:, None])[:, 0].transpose(1, 2).reshape(B, C, Hh, Ww)
    return x + _conv(o, sd, p + 'proj_out.', 0)


def conv_decoder(sd, z, p='superresolution.conv_sr.', num_levels=4, num_res_blocks=1):
    h = _conv(z, sd, p + 'conv_in.', 1)
    h = resnet_block(sd, p + 'mid.block_1.', h)
    h = attn_block(sd, p + 'mid.attn_1.', h)
    h = resnet_block(sd, p + 'mid.block_2.', h)
    for lvl in reversed(range(num_levels)):
        for ib in range(num_res_blocks + 1):
            h = resnet_block(sd, f'{p}up.{lvl}.block.{ib}.', h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')
            h = _conv(h, sd, f'{p}up.{lvl}.upsample.conv.', 1)
    h = _swish(_gn(h, sd, p + 'norm_out.'))
    return _conv(h, sd, p + 'conv_out.', 1)


def vae_decode(sd, latent, num_heads, return_tokens=False):
    """AE.decode_after_vae_no_render: latent [B,12,32,32] -> planes [B,96,128,128]."""
    c = patch_embed_triplane(sd, latent)
    tok = dit2_forward(sd, c, num_heads)
    if return_tokens:
        return tok
    B, L, D = tok.shape
    hw = int((L // 3) ** 0.5)
    z = tok.reshape(B, 3, hw, hw, D).permute(0, 1, 4, 2, 3).reshape(B * 3, D, hw, hw)
    y = conv_decoder(sd, z)                                             # [(b n),32,128,128]
    return y.reshape(B, 3 * y.shape[1], y.shape[2], y.shape[3])
