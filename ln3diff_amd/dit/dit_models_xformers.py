"""Parameter containers + shared HIP execution helpers for the DiT family.

Mirrors the module surface of the reference's dit/dit_models_xformers.py (class names,
constructor arguments, state-dict keys) - the arithmetic is NOT here: the reference's
torch/xformers forward passes are replaced by HIP kernel sequences driven from
`DiTRuntime` (GEMM+epilogue, fused attention, norm+modulate), see DESIGN.md.

state-dict key compatibility: `attn.qkv/proj`, `cross_attn.to_q/to_k/to_v/to_out.0`,
`mlp.mlp.{0,2}.weight`, `mlp.mlp.{1,3}.bias` (xformers FusedMLP naming), `adaLN_modulation.1`,
`norm{1,2}.weight`, `scale_shift_table`, `attn.{q,k}_norm.weight`.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def modulate(x, shift, scale):            # kept for API parity; not used by the HIP path
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def t2i_modulate(x, shift, scale):
    return x * (1 + scale) + shift


# ----------------------------------------------------------------------------- pos embed
def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.)
    omega = 1. / 10000 ** omega
    out = np.einsum('m,d->md', pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0):
    """reference dit/dit_models_xformers.py:961-987 (grid_size int or (h, w) tuple)."""
    if isinstance(grid_size, tuple):
        gh, gw = grid_size
    else:
        gh = gw = grid_size
    grid = np.meshgrid(np.arange(gw, dtype=np.float32), np.arange(gh, dtype=np.float32))
    grid = np.stack(grid, axis=0).reshape([2, 1, gh, gw])
    emb = np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                          get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)
    if cls_token and extra_tokens > 0:
        emb = np.concatenate([np.zeros([extra_tokens, embed_dim]), emb], axis=0)
    return emb


# ----------------------------------------------------------------------------- containers
class _Bias(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(n))


class FusedMLP(nn.Module):
    """Holds xformers-FusedMLP-named parameters: mlp.0.weight, mlp.1.bias, mlp.2.weight, mlp.3.bias."""

    def __init__(self, dim_model, hidden_layer_multiplier=4):
        super().__init__()
        h = hidden_layer_multiplier * dim_model
        self.mlp = nn.Sequential(nn.Linear(dim_model, h, bias=False), _Bias(h),
                                 nn.Linear(h, dim_model, bias=False), _Bias(dim_model))


class RMSNormP(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class Attention(nn.Module):           # vit/vision_transformer.py:59-86 (MemEffAttention container)
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, **_):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.qk_norm = qk_norm
        if qk_norm:
            self.q_norm = RMSNormP(dim // num_heads)
            self.k_norm = RMSNormP(dim // num_heads)


class MemoryEfficientCrossAttention(nn.Module):   # ldm/modules/attention.py:245-276
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, qk_norm=False, **_):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim or query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim))
        self.qk_norm = qk_norm
        if qk_norm:
            self.q_norm = RMSNormP(dim_head)
            self.k_norm = RMSNormP(dim_head)


class Mlp(nn.Module):                  # timm Mlp container (fc1, fc2)
    def __init__(self, in_features, hidden_features, out_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)


class CaptionEmbedder(nn.Module):      # dit_models_xformers.py:183-223
    def __init__(self, in_channels, hidden_size, **_):
        super().__init__()
        self.y_proj = Mlp(in_channels, hidden_size, hidden_size)


class TimestepEmbedder(nn.Module):     # dit_models_xformers.py:87-127
    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size))
        self.frequency_embedding_size = frequency_embedding_size


class PatchEmbed(nn.Module):           # timm PatchEmbed container
    def __init__(self, img_size, patch_size, in_chans, embed_dim, bias=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)


class DiTBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, context_dim=None, enable_rmsnorm=False,
                 norm_type='layernorm', qk_norm=False, **block_kwargs):
        super().__init__()
        self.norm_type = norm_type
        if norm_type == 'rmsnorm':
            self.norm1 = RMSNormP(hidden_size)
            self.norm2 = RMSNormP(hidden_size)
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True, qk_norm=qk_norm)
        self.mlp = FusedMLP(hidden_size, int(mlp_ratio))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))


class TextCondDiTBlock(DiTBlock):      # dit_models_xformers.py:298-323
    def __init__(self, hidden_size, num_heads, mlp_ratio=4, **block_kwargs):
        super().__init__(hidden_size, num_heads, mlp_ratio, **block_kwargs)
        self.cross_attn = MemoryEfficientCrossAttention(query_dim=hidden_size, heads=num_heads)


class ImageCondDiTBlock(DiTBlock):                  # dit_models_xformers.py:417-476
    """Image-conditioned block of the plain DiT_I23D: its OWN adaLN (SiLU -> Linear(D, 6D)), affine-free LayerNorm pre-norms,
    self-attention over [modulated x ; DINO tokens] with qk-norm, cross-attention (qk-norm) over the CLIP tokens normalised by
    the block's attention_y_norm (RMSNorm(1024))."""

    def __init__(self, hidden_size, num_heads, context_dim, mlp_ratio=4, **block_kwargs):
        super().__init__(hidden_size, num_heads, mlp_ratio, context_dim=context_dim, qk_norm=True)
        self.cross_attn = MemoryEfficientCrossAttention(query_dim=hidden_size, context_dim=context_dim, heads=num_heads, qk_norm=True)
        self.attention_y_norm = RMSNormP(1024)


class ImageCondDiTBlockPixelArtRMSNorm(DiTBlock):   # dit_models_xformers.py:481-539,604-618
    def __init__(self, hidden_size, num_heads, context_dim, mlp_ratio=4, **block_kwargs):
        super().__init__(hidden_size, num_heads, mlp_ratio, context_dim=context_dim, norm_type='rmsnorm', qk_norm=True)
        self.cross_attn = MemoryEfficientCrossAttention(query_dim=hidden_size, context_dim=context_dim,
                                                        heads=num_heads, qk_norm=True)
        self.attention_y_norm = RMSNormP(1024)
        self.scale_shift_table = nn.Parameter(torch.randn(6, hidden_size) / hidden_size ** 0.5)
        self.adaLN_modulation = None


class PixelArtTextCondDiTBlock(DiTBlock):          # dit_models_xformers.py:326-369
    """T23D block with PixArt-style conditioning: RMSNorm pre-norms, single shared adaLN + scale_shift_table, cross-attention
    (64-dim heads, no qk-norm) over the text tokens normalised by the block's own attention_y_norm."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4, context_dim=None, **block_kwargs):
        super().__init__(hidden_size, num_heads, mlp_ratio, norm_type='rmsnorm')
        self.cross_attn = MemoryEfficientCrossAttention(query_dim=hidden_size, context_dim=context_dim, heads=num_heads)
        self.scale_shift_table = nn.Parameter(torch.randn(6, hidden_size) / hidden_size ** 0.5)
        self.adaLN_modulation = None
        self.attention_y_norm = RMSNormP(context_dim)


class FinalLayer(nn.Module):           # dit_models_xformers.py:655-678
    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


class T2IFinalLayer(nn.Module):        # dit_models_xformers.py:61-84
    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.scale_shift_table = nn.Parameter(torch.randn(2, hidden_size) / hidden_size ** 0.5)
        self.adaLN_modulation = None
        self.out_channels = out_channels


# ----------------------------------------------------------------------------- packed weights / runtime helpers
def bf16(t, device):
    return t.detach().to(device=device, dtype=torch.bfloat16).contiguous()


def f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Workspace:
    """Shape-keyed cache of device scratch tensors (allocated once; no allocation in the step loop).  One set per network instance:
    a network's forwards are ordered on whatever stream the caller runs them on (the HIP-graph capture warms up and captures on one
    persistent side stream, ordered against the caller's by events); concurrent forwards of ONE instance from several streams are
    not supported (r5 keyed the set by stream handle for an experiment that is gone; it leaked a set per capture stream)."""

    def __init__(self, device):
        self.device = device
        self._t = {}

    def get(self, name, shape, dtype, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._t.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            self._t[key] = t
        return t


def attn_head_pad(Dh):
    """Head size the attention kernel runs at: 64, 80 and 128 natively; smaller heads (the U-Net's 32 / 40) zero-padded to 64, 65 - 80
    (DiT-XL/2: 72, the U-Net's 80) stored 80 wide (r6: the K / V^T stream of such a launch is what bounds it, 80 wide is 5/8 of the
    bytes of 128 wide), anything above to 128."""
    return 64 if Dh <= 64 else (80 if Dh <= 80 else 128)


def attn_out_dim(Dh):
    """Width of one head in the attention OUTPUT: the kernel writes compact heads for the padded sizes it knows (72 -> 72, ABI 8),
    the stored width otherwise."""
    return Dh if Dh in (64, 72, 80, 128) else attn_head_pad(Dh)


def pad_head_columns(w, H, Dh):
    """proj weight [D_out, H*Dh] -> [D_out, H*attn_out_dim] with zero columns, matching the attention output's head width."""
    Dp = attn_out_dim(Dh)
    if Dp == Dh:
        return w
    out = w.new_zeros(w.shape[0], H, Dp)
    out[:, :, :Dh] = w.reshape(w.shape[0], H, Dh)
    return out.reshape(w.shape[0], H * Dp)


def self_attention_hip(ws, tag, h_bf16, B, N, D, H, qkv_w, qkv_b, qn=None, kn=None, nq=None):
    """h [B*N, D] bf16 -> attention output bf16 [B*nq, H*attn_out_dim] (nq <= N query rows kept).  For head sizes other than
    64/128 the QKV epilogue writes into 128-wide zero-initialised heads (exact: the extra dims contribute 0 to q.k); the kernel
    skips the padding for the sizes it knows (DiT-XL/2's 72: compact output, unpadded proj weight), otherwise it produces 0 output
    columns, which meet zero columns of the padded proj weight."""
    Dh = D // H
    Dp = attn_head_pad(Dh)
    nq = N if nq is None else nq
    npad = (N + 63) // 64 * 64
    q = ws.get(tag + 'q', (B, H, npad, Dp), torch.bfloat16, zero=True)
    k = ws.get(tag + 'k', (B, H, npad, Dp), torch.bfloat16, zero=True)
    vt = ws.get(tag + 'vt', (B, H, Dp, npad), torch.bfloat16, zero=True)
    Do = attn_out_dim(Dh)
    o = ws.get(tag + 'o', (B * nq, H * Do), torch.bfloat16)
    fused = qn is not None and ops.heads_norm_fusable(B * N, qkv_w.shape[0], N, Dh, Dp)
    ops.gemm(h_bf16, qkv_w, qkv_b, ops.EPI_HEADS, q, k, vt, M=B * N, tokens=N, tok_pad=npad, heads=H, head_dim=Dh,
             transpose_mask=0b100, head_dim_pad=Dp, head_norm0=qn if fused else None, head_norm1=kn if fused else None)
    if qn is not None and not fused:
        ops.rmsnorm_heads(q, qn, B * H * npad, Dp, true_dim=Dh)      # qn / kn: [Dp] (zero beyond Dh when padded)
        ops.rmsnorm_heads(k, kn, B * H * npad, Dp, true_dim=Dh)
    ops.attention(q, k, vt, o, B, H, nq, npad, N, npad, Dp, scale=Dh ** -0.5, dh_true=Dh if Do != Dp else 0)
    return o


def sincos_timestep_freqs(dim=256, max_period=10000.0):
    half = dim // 2
    return torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
