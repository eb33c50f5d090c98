"""CPU oracle (TEST INFRASTRUCTURE ONLY) - fp32 torch restatement of the reference's
DiT denoisers on the text/image->3D sampling path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (ln3diff_amd/) never does: it runs the HIP kernels or fails loudly.

Pinned against the reference's own Python (run in the build container through
tests/golden/ref_shims.py) by tests/golden/make_golden.py; the committed vectors are
tests/golden/*.npz.  Third-party arithmetic the reference delegates to packages that
are absent from /root/reference (xformers 0.0.26 attention / FusedMLP, timm 0.6
PatchEmbed / Mlp) is restated from its published semantics - "parity unpinned"
against those packages themselves.

All functions take `sd`, a state-dict with the reference's parameter names, so the
same weights drive the reference module, this oracle and the HIP product modules.

Reference citations (relative to the reference checkout):
  timestep_embedding   dit/dit_models_xformers.py:101-127
  pos embed            dit/dit_models_xformers.py:961-1021, dit/dit_trilatent.py:51-66
  modulate             dit/dit_models_xformers.py:48-53
  self attention       vit/vision_transformer.py:108-124 (MemEffAttention)
  cross attention      ldm/modules/attention.py:278-307
  RMSNorm              dit/norm.py:27-40
  TextCondDiTBlock     dit/dit_models_xformers.py:306-323
  ImageCond..PixelArt  dit/dit_models_xformers.py:506-539
  FinalLayer/T2IFinal  dit/dit_models_xformers.py:655-678 / :61-84
  DiT_TriLatent.fwd    dit/dit_trilatent.py:74-143
  DiT_I23D_PixelArt    dit/dit_i23d.py:229-291, forward_with_cfg :155-168
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

DIT_CONFIGS = {
    # arch: (hidden, depth, heads)   reference dit/dit_trilatent.py:272-293
    "DiT-B/2": (768, 12, 12),
    "DiT-L/2": (1024, 24, 16),
    "DiT-XL/2": (1152, 28, 16),
    "DiT-B/1": (768, 12, 12),        # patch size 1 (pass patch=1 to t23d_forward)
}


# ------------------------------------------------------------------ embeddings
def timestep_embedding(t, dim=256, max_period=10000.0):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) *
                      torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed_2d(embed_dim, grid_hw):
    """get_2d_sincos_pos_embed with a (h, w) tuple: first half of the channels encodes
    grid[0] (the w-index, 'w goes first'), second half grid[1] (the h-index)."""
    gh, gw = grid_hw
    grid_h = np.arange(gh, dtype=np.float32)
    grid_w = np.arange(gw, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape(2, 1, gh, gw)
    emb_h = _sincos_1d(embed_dim // 2, grid[0])
    emb_w = _sincos_1d(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


def trilatent_pos_embed(D, plane_n=3, tokens_per_plane=256):
    """init_PE_3D_aware: grid (plane_n, p*p)."""
    pe = sincos_pos_embed_2d(D, (plane_n, tokens_per_plane))
    return torch.from_numpy(pe).float().reshape(1, plane_n * tokens_per_plane, D)


# ----------------------------------------------------------------- primitives
# Operand-rounding emulation (test infrastructure for the precision argument of DESIGN.md): with OPERAND_ROUND[0] = torch.bfloat16
# every GEMM operand (activations, weights, attention q/k/v and probabilities) is rounded to bf16 and the product accumulated
# in fp32 - the arithmetic of the HIP path's MFMA kernels.  The HIP forward must agree with THIS restatement an order of
# magnitude more tightly than with the fp32 one; what is left is accumulation order and rounding-boundary flips.
OPERAND_ROUND = [None]


class operand_rounding:
    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype

    def __enter__(self):
        self.prev, OPERAND_ROUND[0] = OPERAND_ROUND[0], self.dtype

    def __exit__(self, *a):
        OPERAND_ROUND[0] = self.prev


def _r(x):
    return x if OPERAND_ROUND[0] is None or x is None else x.to(OPERAND_ROUND[0]).to(x.dtype)


def linear(x, w, b=None):
    return F.linear(_r(x), _r(w), b)


def layer_norm(x, eps=1e-6, w=None, b=None):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def rms_norm(x, w, eps=1e-5):
    var = x.float().pow(2).mean(-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    return x * w if w is not None else x


def sdpa(q, k, v):
    """q,k,v [B,H,N,Dh]: softmax(q k^T / sqrt(Dh)) v, no mask, no dropout."""
    if OPERAND_ROUND[0] is None:
        s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
        return torch.softmax(s, dim=-1) @ v
    # the kernels' arithmetic: q pre-scaled then rounded, un-normalised probabilities rounded for the P.V product, fp32 row sum
    s = _r(q * (q.shape[-1] ** -0.5)) @ _r(k).transpose(-1, -2)
    p = torch.exp(s - s.amax(-1, keepdim=True))
    return (_r(p) @ _r(v)) / p.sum(-1, keepdim=True)


def self_attention(sd, p, x, H):
    B, N, C = x.shape
    qkv = linear(x, sd[p + 'qkv.weight'], sd[p + 'qkv.bias']).reshape(B, N, 3, H, C // H)
    q, k, v = qkv.unbind(2)                                  # [B,N,H,Dh]
    if p + 'q_norm.weight' in sd:
        q = rms_norm(q, sd[p + 'q_norm.weight'])
        k = rms_norm(k, sd[p + 'k_norm.weight'])
    o = sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    o = o.transpose(1, 2).reshape(B, N, C)
    return linear(o, sd[p + 'proj.weight'], sd[p + 'proj.bias'])


def cross_attention(sd, p, x, ctx, H, dim_head=64):
    B = x.shape[0]
    q = linear(x, sd[p + 'to_q.weight'])
    k = linear(ctx, sd[p + 'to_k.weight'])
    v = linear(ctx, sd[p + 'to_v.weight'])
    sp = lambda t: t.reshape(B, t.shape[1], H, dim_head).transpose(1, 2)
    q, k, v = sp(q), sp(k), sp(v)
    if p + 'q_norm.weight' in sd:
        q = rms_norm(q, sd[p + 'q_norm.weight'])
        k = rms_norm(k, sd[p + 'k_norm.weight'])
    o = sdpa(q, k, v).transpose(1, 2).reshape(B, x.shape[1], H * dim_head)
    return linear(o, sd[p + 'to_out.0.weight'], sd[p + 'to_out.0.bias'])


def fused_mlp(sd, p, x):
    h = linear(x, sd[p + 'mlp.0.weight']) + sd[p + 'mlp.1.bias']
    h = F.gelu(h)                                            # erf GELU
    return linear(h, sd[p + 'mlp.2.weight']) + sd[p + 'mlp.3.bias']


def caption_embedder(sd, p, c):
    h = linear(c, sd[p + 'y_proj.fc1.weight'], sd[p + 'y_proj.fc1.bias'])
    h = F.gelu(h, approximate='tanh')
    return linear(h, sd[p + 'y_proj.fc2.weight'], sd[p + 'y_proj.fc2.bias'])


def t_embedder(sd, t):
    h = linear(timestep_embedding(t), sd['t_embedder.mlp.0.weight'], sd['t_embedder.mlp.0.bias'])
    return linear(F.silu(h), sd['t_embedder.mlp.2.weight'], sd['t_embedder.mlp.2.bias'])


def patchify_embed(sd, x, patch=2):
    """'b (c n) h w -> (b n) c h w', shared conv patch-embed, '(b n) l c -> b (n l) c'."""
    B, C3, Hh, Ww = x.shape
    C = C3 // 3
    xp = x.reshape(B, C, 3, Hh, Ww).permute(0, 2, 1, 3, 4).reshape(B * 3, C, Hh, Ww)
    tok = F.conv2d(xp, sd['x_embedder.proj.weight'], sd['x_embedder.proj.bias'], stride=patch)
    D = tok.shape[1]
    tok = tok.flatten(2).transpose(1, 2)                      # [(b n), l, D]
    return tok.reshape(B, 3 * tok.shape[1], D)


def unpatchify_trilatent(y, B, patch, c_out):
    """[B, 3*l, p*p*c] -> [B, c*3, H, W] with channel index c*3+n."""
    l = y.shape[1] // 3
    h = w = int(l ** 0.5)
    y = y.reshape(B * 3, h, w, patch, patch, c_out)
    y = torch.einsum('nhwpqc->nchpwq', y).reshape(B * 3, c_out, h * patch, w * patch)
    y = y.reshape(B, 3, c_out, h * patch, w * patch).permute(0, 2, 1, 3, 4)
    return y.reshape(B, c_out * 3, h * patch, w * patch).contiguous()


# --------------------------------------------------------------------- T23D
def t23d_block(sd, p, x, t_emb, ctx, H):
    mod = linear(F.silu(t_emb), sd[p + 'adaLN_modulation.1.weight'], sd[p + 'adaLN_modulation.1.bias'])
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    h = layer_norm(x) * (1 + sc_a[:, None]) + sh_a[:, None]
    x = x + g_a[:, None] * self_attention(sd, p + 'attn.', h, H)
    x = x + cross_attention(sd, p + 'cross_attn.', x, ctx, H)
    h = layer_norm(x) * (1 + sc_m[:, None]) + sh_m[:, None]
    return x + g_m[:, None] * fused_mlp(sd, p + 'mlp.', h)


def t23d_forward(sd, x, timesteps, context, num_heads, patch=2, return_tokens=False):
    """DiT_TriLatent.forward with vit_blk=TextCondDiTBlock, FinalLayer."""
    if isinstance(context, dict):
        context = context['crossattn']
    B = x.shape[0]
    depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    t_emb = t_embedder(sd, timesteps)
    h = patchify_embed(sd, x, patch) + sd['pos_embed']
    ctx = caption_embedder(sd, 'clip_text_proj.', context)
    for i in range(depth):
        h = t23d_block(sd, f'blocks.{i}.', h, t_emb, ctx, num_heads)
    if return_tokens:
        return h
    mod = linear(F.silu(t_emb), sd['final_layer.adaLN_modulation.1.weight'],
                   sd['final_layer.adaLN_modulation.1.bias'])
    shift, scale = mod.chunk(2, dim=1)
    y = layer_norm(h) * (1 + scale[:, None]) + shift[:, None]
    y = F.linear(y, sd['final_layer.linear.weight'], sd['final_layer.linear.bias'])     # fp32 in the HIP path too (final_layer kernel)
    c_out = y.shape[-1] // (patch * patch)
    return unpatchify_trilatent(y, B, patch, c_out).float()


def t23d_pixart_forward(sd, x, timesteps, context, num_heads, patch=2):
    """DiT_TriLatent_PixelArt.forward (dit/dit_trilatent.py:146-250; registry 'DiT-PixelArt-L/2' / '-B/2') with
    PixelArtTextCondDiTBlock (dit_models_xformers.py:326-369): t = t_embedder + cap_embedder(LN -> Linear)(context['vector']), ONE
    shared adaLN whose output is added to each block's scale_shift_table, RMSNorm (affine, eps 1e-5) pre-norms, self-attention without
    qk-norm, cross-attention (64-dim heads, no qk-norm) over the text tokens normalised by the BLOCK's attention_y_norm, T2IFinalLayer."""
    B = x.shape[0]
    depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    vec = context['vector'].float()
    cls = linear(layer_norm(vec, 1e-5, sd['cap_embedder.0.weight'], sd['cap_embedder.0.bias']),
                 sd['cap_embedder.1.weight'], sd['cap_embedder.1.bias'])
    ctx = context['crossattn'].float()
    t = t_embedder(sd, timesteps.float()) + cls
    t0 = linear(F.silu(t), sd['adaLN_modulation.1.weight'], sd['adaLN_modulation.1.bias'])
    h = patchify_embed(sd, x, patch) + sd['pos_embed']
    for i in range(depth):
        p = f'blocks.{i}.'
        mod = sd[p + 'scale_shift_table'][None] + t0.reshape(B, 6, -1)
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
        h = h + g_a * self_attention(sd, p + 'attn.', rms_norm(h, sd[p + 'norm1.weight']) * (1 + sc_a) + sh_a, num_heads)
        h = h + cross_attention(sd, p + 'cross_attn.', h, rms_norm(ctx, sd[p + 'attention_y_norm.weight']), num_heads)
        h = h + g_m * fused_mlp(sd, p + 'mlp.', rms_norm(h, sd[p + 'norm2.weight']) * (1 + sc_m) + sh_m)
    shift, scale = (sd['final_layer.scale_shift_table'][None] + t[:, None]).chunk(2, dim=1)
    y = layer_norm(h) * (1 + scale) + shift
    y = F.linear(y, sd['final_layer.linear.weight'], sd['final_layer.linear.bias'])
    c_out = y.shape[-1] // (patch * patch)
    return unpatchify_trilatent(y, B, patch, c_out).float()


# --------------------------------------------------------------------- I23D
def i23d_block(sd, p, x, t0, dino_tok, clip_tok, H):
    B, N, D = x.shape
    mod = sd[p + 'scale_shift_table'][None] + t0.reshape(B, 6, -1)
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)     # each [B,1,D]
    h = rms_norm(x, sd[p + 'norm1.weight']) * (1 + sc_a) + sh_a
    h = torch.cat([h, dino_tok], dim=1)
    x = x + g_a * self_attention(sd, p + 'attn.', h, H)[:, :N]
    x = x + cross_attention(sd, p + 'cross_attn.', x, clip_tok, H)
    h = rms_norm(x, sd[p + 'norm2.weight']) * (1 + sc_m) + sh_m
    return x + g_m * fused_mlp(sd, p + 'mlp.', h)


def i23d_forward(sd, x, timesteps, context, num_heads, patch=2, clip_ctx_dim=1024):
    """DiT_I23D_PixelArt.forward (ImageCondDiTBlockPixelArtRMSNorm blocks, T2IFinalLayer)."""
    B = x.shape[0]
    depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    vec = context['vector'].float()
    cls = linear(layer_norm(vec, 1e-5, sd['cap_embedder.0.weight'], sd['cap_embedder.0.bias']),
                   sd['cap_embedder.1.weight'], sd['cap_embedder.1.bias'])
    ca = context['crossattn'].float()
    clip_tok = rms_norm(ca[..., :clip_ctx_dim], sd['attention_y_norm.weight'])
    dino_tok = caption_embedder(sd, 'dino_proj.', ca[..., clip_ctx_dim:])
    t = t_embedder(sd, timesteps.float()) + cls
    t0 = linear(F.silu(t), sd['adaLN_modulation.1.weight'], sd['adaLN_modulation.1.bias'])
    h = patchify_embed(sd, x, patch) + sd['pos_embed']
    for i in range(depth):
        h = i23d_block(sd, f'blocks.{i}.', h, t0, dino_tok, clip_tok, num_heads)
    shift, scale = (sd['final_layer.scale_shift_table'][None] + t[:, None]).chunk(2, dim=1)
    y = layer_norm(h) * (1 + scale) + shift
    y = F.linear(y, sd['final_layer.linear.weight'], sd['final_layer.linear.bias'])     # fp32 in the HIP path too (final_layer kernel)
    c_out = y.shape[-1] // (patch * patch)
    return unpatchify_trilatent(y, B, patch, c_out).float()


def i23d_plain_block(sd, p, x, t_emb, dino_tok, clip_tok, H):
    """ImageCondDiTBlock.forward (dit/dit_models_xformers.py:450-476): the block's own adaLN, affine-free LayerNorm pre-norms,
    [modulated x ; DINO] self-attention, cross-attention over the block's attention_y_norm(CLIP tokens)."""
    N = x.shape[1]
    mod = linear(F.silu(t_emb), sd[p + 'adaLN_modulation.1.weight'], sd[p + 'adaLN_modulation.1.bias'])
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    h = layer_norm(x) * (1 + sc_a[:, None]) + sh_a[:, None]
    h = torch.cat([h, dino_tok], dim=1)
    x = x + g_a[:, None] * self_attention(sd, p + 'attn.', h, H)[:, :N]
    x = x + cross_attention(sd, p + 'cross_attn.', x, rms_norm(clip_tok, sd[p + 'attention_y_norm.weight']), H)
    h = layer_norm(x) * (1 + sc_m[:, None]) + sh_m[:, None]
    return x + g_m[:, None] * fused_mlp(sd, p + 'mlp.', h)


def i23d_plain_forward(sd, x, timesteps, context, num_heads, patch=2, clip_ctx_dim=1024):
    """DiT_I23D.forward (dit/dit_i23d.py:96-153): t = t_embedder + clip_text_proj(context['vector']); T2IFinalLayer."""
    B = x.shape[0]
    depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    ca = context['crossattn'].float()
    cls = caption_embedder(sd, 'clip_text_proj.', context['vector'].float())
    clip_tok, dino_tok = ca[..., :clip_ctx_dim], caption_embedder(sd, 'dino_proj.', ca[..., clip_ctx_dim:])
    t = t_embedder(sd, timesteps.float()) + cls
    h = patchify_embed(sd, x, patch) + sd['pos_embed']
    for i in range(depth):
        h = i23d_plain_block(sd, f'blocks.{i}.', h, t, dino_tok, clip_tok, num_heads)
    shift, scale = (sd['final_layer.scale_shift_table'][None] + t[:, None]).chunk(2, dim=1)
    y = layer_norm(h) * (1 + scale) + shift
    y = F.linear(y, sd['final_layer.linear.weight'], sd['final_layer.linear.bias'])
    return unpatchify_trilatent(y, B, patch, y.shape[-1] // (patch * patch)).float()


def i23d_mv_forward(sd, x, timesteps, context, num_heads, patch=2):
    """DiT_I23D_PixelArt_MVCond.forward (dit/dit_i23d.py:293-384): multi-view image conditioning.  The projected CLIP spatial
    tokens (clip_spatial_proj) are the ones appended to the self-attention sequence and the flattened multi-view DINO features
    context['concat'] [B, V, L, C] are the cross-attention context, raw (no attention_y_norm, no projection); dino_proj is gone."""
    B = x.shape[0]
    depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    vec = context['vector'].float()
    cls = linear(layer_norm(vec, 1e-5, sd['cap_embedder.0.weight'], sd['cap_embedder.0.bias']),
                   sd['cap_embedder.1.weight'], sd['cap_embedder.1.bias'])
    clip_tok = caption_embedder(sd, 'clip_spatial_proj.', context['crossattn'].float())
    mv = context['concat'].float()
    dino_tok = mv.reshape(B, mv.shape[1] * mv.shape[2], mv.shape[3])
    t = t_embedder(sd, timesteps.float()) + cls
    t0 = linear(F.silu(t), sd['adaLN_modulation.1.weight'], sd['adaLN_modulation.1.bias'])
    h = patchify_embed(sd, x, patch) + sd['pos_embed']
    for i in range(depth):
        h = i23d_block(sd, f'blocks.{i}.', h, t0, clip_tok, dino_tok, num_heads)       # (appended, cross-attended)
    shift, scale = (sd['final_layer.scale_shift_table'][None] + t[:, None]).chunk(2, dim=1)
    y = layer_norm(h) * (1 + scale) + shift
    y = F.linear(y, sd['final_layer.linear.weight'], sd['final_layer.linear.bias'])     # fp32 in the HIP path too (final_layer kernel)
    c_out = y.shape[-1] // (patch * patch)
    return unpatchify_trilatent(y, B, patch, c_out).float()


def i23d_pcd_forward(sd, x, timesteps, context, num_heads):
    """DiT_pcd_I23D_PixelArt_MVCond.forward (dit/dit_i23d.py:500-588, registry key 'DiT-PixArt-MV-PCD-L'): the tokens are the
    points of a point-cloud latent x [B, N, C] - no patchify, no positional embedding; x_embedder is a timm Mlp
    (Linear -> tanh-GELU -> Linear); conditioning and blocks as MVCond; the output stays [B, N, out_channels]."""
    B = x.shape[0]
    depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    vec = context['vector'].float()
    cls = linear(layer_norm(vec, 1e-5, sd['cap_embedder.0.weight'], sd['cap_embedder.0.bias']),
                 sd['cap_embedder.1.weight'], sd['cap_embedder.1.bias'])
    clip_tok = caption_embedder(sd, 'clip_spatial_proj.', context['crossattn'].float())
    mv = context['concat'].float()
    dino_tok = mv.reshape(B, mv.shape[1] * mv.shape[2], mv.shape[3])
    t = t_embedder(sd, timesteps.float()) + cls
    t0 = linear(F.silu(t), sd['adaLN_modulation.1.weight'], sd['adaLN_modulation.1.bias'])
    h = linear(x.float(), sd['x_embedder.fc1.weight'], sd['x_embedder.fc1.bias'])
    h = linear(F.gelu(h, approximate='tanh'), sd['x_embedder.fc2.weight'], sd['x_embedder.fc2.bias'])
    for i in range(depth):
        h = i23d_block(sd, f'blocks.{i}.', h, t0, clip_tok, dino_tok, num_heads)
    shift, scale = (sd['final_layer.scale_shift_table'][None] + t[:, None]).chunk(2, dim=1)
    y = layer_norm(h) * (1 + scale) + shift
    return linear(y, sd['final_layer.linear.weight'], sd['final_layer.linear.bias']).float()


def i23d_mv_noclip_forward(sd, x, timesteps, context, num_heads, patch=2):
    """DiT_I23D_PixelArt_MVCond_noClip.forward (dit/dit_i23d.py:387-492, the class the reference registers as
    'DiT-PixArt-MV-L/2'): no CLIP branch at all - t = t_embedder(timesteps), nothing is appended to the self-attention
    sequence (ImageCondDiTBlockPixelArtNoclip, dit_models_xformers.py:540-596), cross-attention over the flattened
    multi-view DINO features."""
    B = x.shape[0]
    depth = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    mv = context['concat'].float()
    dino_tok = mv.reshape(B, mv.shape[1] * mv.shape[2], mv.shape[3])
    t = t_embedder(sd, timesteps.float())
    t0 = linear(F.silu(t), sd['adaLN_modulation.1.weight'], sd['adaLN_modulation.1.bias'])
    h = patchify_embed(sd, x, patch) + sd['pos_embed']
    none = h.new_zeros(B, 0, h.shape[-1])
    for i in range(depth):
        h = i23d_block(sd, f'blocks.{i}.', h, t0, none, dino_tok, num_heads)
    shift, scale = (sd['final_layer.scale_shift_table'][None] + t[:, None]).chunk(2, dim=1)
    y = layer_norm(h) * (1 + scale) + shift
    y = F.linear(y, sd['final_layer.linear.weight'], sd['final_layer.linear.bias'])     # fp32 in the HIP path too (final_layer kernel)
    c_out = y.shape[-1] // (patch * patch)
    return unpatchify_trilatent(y, B, patch, c_out).float()


def i23d_forward_with_cfg(sd, x, t, context, cfg_scale, num_heads):
    eps = i23d_forward(sd, x, t, context, num_heads)
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half = uncond + cfg_scale * (cond - uncond)
    return torch.cat([half, half], dim=0)
