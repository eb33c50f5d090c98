"""CPU oracle (TEST INFRASTRUCTURE ONLY) - fp32 torch restatement of the U-Net denoiser behind the ShapeNet / FFHQ entry point
(scripts/vit_triplane_diffusion_sample.py).  Never imported by the product path.

Reference citations:
  UNetModel               guided_diffusion/unet.py:427-791 (constructor wiring :553-726, forward :752-791)
  ResBlock                guided_diffusion/unet.py:164-278 (use_scale_shift_norm branch :267-271)
  Downsample / Upsample   guided_diffusion/unet.py:102-161
  AttentionBlock          guided_diffusion/unet.py:281-336, QKVAttentionLegacy :359-389 (the non-transformer variant)
  SpatialTransformer      ldm/modules/attention_compat.py:228-277 (GroupNorm eps 1e-6, 1x1 proj_in / proj_out)
  BasicTransformerBlock   ldm/modules/attention_compat.py:205-225, CrossAttention :161-202, FeedForward / GEGLU :45-82
  timestep_embedding      guided_diffusion/nn.py:103-121, normalization = GroupNorm32(32, C) (eps 1e-5) :93-100
  mixed prediction        guided_diffusion/gaussian_diffusion.py:327-348, :548-558; continuous_diffusion_utils.py:748-754

`layout(cfg)` replays the constructor's wiring and returns, per block of input_blocks / middle_block / output_blocks, the list of
layer kinds - the same walk the product's ln3diff_amd.guided_diffusion.unet does.
"""
import math

import torch
import torch.nn.functional as F


def default_channel_mult(image_size):
    """guided_diffusion/script_util.py:292-317"""
    return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 4, 4),
            16: (1, 2, 3, 4)}[image_size]


def layout(cfg):
    """cfg: image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions (ds values), channel_mult,
    num_heads, num_head_channels (-1), use_spatial_transformer, transformer_depth, context_dim, legacy (True), resblock_updown (False).
    Returns (input_blocks, middle_block, output_blocks): lists of blocks, a block = list of (kind, params)."""
    mc, nrb, cm = cfg['model_channels'], cfg['num_res_blocks'], cfg['channel_mult']
    nh, nhc = cfg['num_heads'], cfg.get('num_head_channels', -1)
    st, legacy = cfg['use_spatial_transformer'], cfg.get('legacy', True)
    assert not cfg.get('resblock_updown', False)

    def attn(ch):
        heads = nh
        if nhc == -1:
            dim_head = ch // heads
        else:
            heads = ch // nhc
            dim_head = nhc
        if legacy:
            dim_head = ch // heads if st else nhc
        if st:
            return ('transformer', dict(ch=ch, heads=heads, dim_head=dim_head, depth=cfg.get('transformer_depth', 1),
                                        context_dim=cfg['context_dim']))
        return ('attention', dict(ch=ch, heads=heads if dim_head == -1 else ch // dim_head))

    inp = [[('conv', dict(cin=cfg['in_channels'], cout=mc))]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cm):
        for _ in range(nrb):
            layers = [('res', dict(cin=ch, cout=int(mult * mc)))]
            ch = int(mult * mc)
            if ds in cfg['attention_resolutions']:
                layers.append(attn(ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(cm) - 1:
            inp.append([('down', dict(ch=ch))])
            chans.append(ch)
            ds *= 2
    mid = [('res', dict(cin=ch, cout=ch)), attn(ch), ('res', dict(cin=ch, cout=ch))]
    out = []
    for level, mult in list(enumerate(cm))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [('res', dict(cin=ch + ich, cout=int(mc * mult)))]
            ch = int(mc * mult)
            if ds in cfg['attention_resolutions']:
                layers.append(attn(ch))
            if level and i == nrb:
                layers.append(('up', dict(ch=ch)))
                ds //= 2
            out.append(layers)
    return inp, mid, out


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, pre, x, eps):
    return F.group_norm(x.float(), 32, sd[pre + '.weight'], sd[pre + '.bias'], eps)


def _conv(sd, pre, x, stride=1, padding=1):
    return F.conv2d(x, sd[pre + '.weight'], sd[pre + '.bias'], stride=stride, padding=padding)


def res_block(sd, pre, x, emb, scale_shift):
    h = _conv(sd, pre + '.in_layers.2', F.silu(_gn(sd, pre + '.in_layers.0', x, 1e-5)))
    e = F.linear(F.silu(emb), sd[pre + '.emb_layers.1.weight'], sd[pre + '.emb_layers.1.bias'])[:, :, None, None]
    if scale_shift:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = _gn(sd, pre + '.out_layers.0', h, 1e-5) * (1 + scale) + shift
        h = _conv(sd, pre + '.out_layers.3', F.silu(h))
    else:
        h = _conv(sd, pre + '.out_layers.3', F.silu(_gn(sd, pre + '.out_layers.0', h + e, 1e-5)))
    skip = x if (pre + '.skip_connection.weight') not in sd else _conv(sd, pre + '.skip_connection', x, padding=0)
    return skip + h


def cross_attention(sd, pre, x, context, heads):
    """CrossAttention.forward: x [B,N,D], context [B,L,Dc] or None (self-attention)."""
    ctx = x if context is None else context
    q = F.linear(x, sd[pre + '.to_q.weight'])
    k = F.linear(ctx, sd[pre + '.to_k.weight'])
    v = F.linear(ctx, sd[pre + '.to_v.weight'])
    B, N, inner = q.shape
    dh = inner // heads
    sp = lambda t: t.reshape(B, -1, heads, dh).permute(0, 2, 1, 3)
    sim = torch.einsum('bhid,bhjd->bhij', sp(q), sp(k)) * dh ** -0.5
    o = torch.einsum('bhij,bhjd->bhid', sim.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(B, N, inner)
    return F.linear(o, sd[pre + '.to_out.0.weight'], sd[pre + '.to_out.0.bias'])


def spatial_transformer(sd, pre, x, context, p):
    B, C, H, W = x.shape
    h = _conv(sd, pre + '.proj_in', _gn(sd, pre + '.norm', x, 1e-6), padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, -1)
    D = h.shape[-1]
    for d in range(p['depth']):
        b = f'{pre}.transformer_blocks.{d}'
        ln = lambda n, t: F.layer_norm(t, (D,), sd[f'{b}.{n}.weight'], sd[f'{b}.{n}.bias'], 1e-5)
        h = cross_attention(sd, b + '.attn1', ln('norm1', h), None, p['heads']) + h
        h = cross_attention(sd, b + '.attn2', ln('norm2', h), context, p['heads']) + h
        g = F.linear(ln('norm3', h), sd[b + '.ff.net.0.proj.weight'], sd[b + '.ff.net.0.proj.bias'])
        a, gate = g.chunk(2, dim=-1)
        h = F.linear(a * F.gelu(gate), sd[b + '.ff.net.2.weight'], sd[b + '.ff.net.2.bias']) + h
    h = h.reshape(B, H, W, D).permute(0, 3, 1, 2)
    return _conv(sd, pre + '.proj_out', h, padding=0) + x


def attention_block(sd, pre, x, p):
    """AttentionBlock + QKVAttentionLegacy (heads split before q / k / v)."""
    B, C, H, W = x.shape
    xf = x.reshape(B, C, -1)
    qkv = F.conv1d(F.group_norm(xf.float(), 32, sd[pre + '.norm.weight'], sd[pre + '.norm.bias'], 1e-5), sd[pre + '.qkv.weight'],
                   sd[pre + '.qkv.bias'])
    nh = p['heads']
    ch = C // nh
    q, k, v = qkv.reshape(B * nh, ch * 3, -1).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum('bct,bcs->bts', q * s, k * s).softmax(-1)
    a = torch.einsum('bts,bcs->bct', w, v).reshape(B, -1, H * W)
    return (xf + F.conv1d(a, sd[pre + '.proj_out.weight'], sd[pre + '.proj_out.bias'])).reshape(B, C, H, W)


def unet_forward(sd, cfg, x, timesteps, context=None):
    """UNetModel.forward.  x [B, C, S, S] (roll_out: [B, 3C, S, S] re-laid as [B, C, S, 3S] like :775-776)."""
    if isinstance(context, dict):
        context = context['crossattn']
    inp, mid, out = layout(cfg)
    ss = cfg.get('use_scale_shift_norm', True)
    emb = timestep_embedding(timesteps, cfg['model_channels'])
    emb = F.linear(F.silu(F.linear(emb, sd['time_embed.0.weight'], sd['time_embed.0.bias'])), sd['time_embed.2.weight'], sd['time_embed.2.bias'])
    if cfg.get('roll_out', False):
        B, C3, H, W = x.shape
        x = x.reshape(B, 3, C3 // 3, H, W).permute(0, 2, 3, 1, 4).reshape(B, C3 // 3, H, 3 * W)

    def run(prefix, layers, h):
        for li, (kind, p) in enumerate(layers):
            pre = f'{prefix}.{li}'
            if kind == 'conv':
                h = _conv(sd, pre, h)
            elif kind == 'res':
                h = res_block(sd, pre, h, emb, ss)
            elif kind == 'transformer':
                h = spatial_transformer(sd, pre, h, context, p)
            elif kind == 'attention':
                h = attention_block(sd, pre, h, p)
            elif kind == 'down':
                h = _conv(sd, pre + '.op', h, stride=2)
            elif kind == 'up':
                h = _conv(sd, pre + '.conv', F.interpolate(h, scale_factor=2, mode='nearest'))
        return h

    hs = []
    h = x.float()
    for bi, layers in enumerate(inp):
        h = run(f'input_blocks.{bi}', layers, h)
        hs.append(h)
    h = run('middle_block', mid, h)
    for bi, layers in enumerate(out):
        h = run(f'output_blocks.{bi}', layers, torch.cat([h, hs.pop()], dim=1))
    h = _conv(sd, 'out.2', F.silu(_gn(sd, 'out.0', h, 1e-5)))
    if cfg.get('roll_out', False):
        B, C, H, W3 = h.shape
        h = h.reshape(B, C, H, 3, W3 // 3).permute(0, 3, 1, 2, 4).reshape(B, 3 * C, H, W3 // 3)
    return h


def mixed_prediction(eps, x, mixing_logit, sqrt_one_minus_ab):
    """get_mixed_prediction on an eps model (gaussian_diffusion.py:336-348): (1 - s) * sqrt(1 - ab_t) * x_t + s * eps, s = sigmoid(logit)."""
    c = torch.sigmoid(mixing_logit)
    return (1 - c) * (sqrt_one_minus_ab * x) + c * eps
