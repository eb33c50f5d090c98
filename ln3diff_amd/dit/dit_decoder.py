"""DiT2 - the conditional ViT used as the tri-plane VAE decoder backbone (reference dit/dit_decoder.py:19-151).

Per-token adaLN (every token carries its own conditioning c), blocks alternate in-plane attention
('(b n) l c', 3B x 256 tokens) and global attention (B x 768).  HIP execution: per block ONE GEMM produces the
per-token modulation [tokens, 6D] from silu(c) (bf16, emitted by the tokeniser kernel), then the same
norm+modulate / QKV-heads GEMM / fused attention / gate+residual GEMM sequence as the denoiser with
per-row gates (gate_rows = 1)."""
import torch
import torch.nn as nn

from .. import ops
from .. import _cache
from .dit_models_xformers import (DiTBlock, Workspace, bf16, f32, get_2d_sincos_pos_embed, self_attention_hip,
                                  pad_head_columns)


class DiTBlock2(DiTBlock):
    pass


class DiT2(nn.Module):
    def __init__(self, input_size=16, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16,
                 mlp_ratio=4, class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, mixing_logit_init=-3,
                 mixed_prediction=True, context_dim=False, roll_out=False, plane_n=3, return_all_layers=False,
                 vit_blk=...):
        super().__init__()
        self.embed_dim, self.depth, self.num_heads = hidden_size, depth, num_heads
        self.roll_out, self.plane_n = roll_out, plane_n
        L = (input_size // patch_size) ** 2 if input_size != 16 else 256
        self.pos_embed = nn.Parameter(torch.zeros(1, plane_n * 256, hidden_size), requires_grad=False)
        self.blocks = nn.ModuleList([DiTBlock2(hidden_size, num_heads, mlp_ratio=mlp_ratio) for _ in range(depth)])
        pe = get_2d_sincos_pos_embed(hidden_size, (3 * 16, 16)).reshape(3 * 256, hidden_size)   # vit_triplane.py:333-343
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        self._packed = None
        _cache.watch(self)

    def _apply(self, fn, *a, **k):
        _cache.bump()
        return super()._apply(fn, *a, **k)

    def pack(self, device):
        if _cache.fresh(self._packed, device):
            return self._packed
        P = {'device': device, 'pos': f32(self.pos_embed[0], device), 'blocks': []}
        for b in self.blocks:
            q = {'ada_w': bf16(b.adaLN_modulation[1].weight, device), 'ada_b': f32(b.adaLN_modulation[1].bias, device),
                 'qkv_w': bf16(b.attn.qkv.weight, device), 'qkv_b': f32(b.attn.qkv.bias, device),
                 'proj_w': bf16(pad_head_columns(b.attn.proj.weight.detach(), self.num_heads, self.embed_dim // self.num_heads), device), 'proj_b': f32(b.attn.proj.bias, device),
                 'fc1_w': bf16(b.mlp.mlp[0].weight, device), 'fc1_b': f32(b.mlp.mlp[1].bias, device),
                 'fc2_w': bf16(b.mlp.mlp[2].weight, device), 'fc2_b': f32(b.mlp.mlp[3].bias, device)}
            P['blocks'].append(q)
        self._packed = _cache.stamp(P, self)
        return P

    @torch.no_grad()
    def forward_tokens(self, silu_c, B, ws):
        """silu_c bf16 [B*768, D] -> tokens f32 [B*768, D] (workspace tensor)."""
        dev = silu_c.device
        P = self.pack(dev)
        D, H, N = self.embed_dim, self.num_heads, self.plane_n * 256
        M = B * N
        x = ws.get('d2_x', (M, D), torch.float32)
        ops.tile_rows(P['pos'], x, N * D, B)
        mod = ws.get('d2_mod', (M, 6 * D), torch.float32)
        hb = ws.get('d2_h', (M, D), torch.bfloat16)
        f1 = ws.get('d2_f1', (M, P['blocks'][0]['fc1_w'].shape[0]), torch.bfloat16)
        for i, q in enumerate(P['blocks']):
            ops.gemm(silu_c, q['ada_w'], q['ada_b'], ops.EPI_F32, mod)
            sh_a, sc_a, g_a = mod[:, 0:], mod[:, D:], mod[:, 2 * D:]
            sh_m, sc_m, g_m = mod[:, 3 * D:], mod[:, 4 * D:], mod[:, 5 * D:]
            ops.norm_modulate(x, hb, M, D, kind=0, eps=1e-6, shift=sh_a, scale=sc_a, mod_rows=1, mod_ld=6 * D)
            if self.roll_out and i % 2 == 0:
                ao = self_attention_hip(ws, 'd2p_', hb, B * self.plane_n, 256, D, H, q['qkv_w'], q['qkv_b'])
            else:
                ao = self_attention_hip(ws, 'd2g_', hb, B, N, D, H, q['qkv_w'], q['qkv_b'])
            ops.gemm(ao, q['proj_w'], q['proj_b'], ops.EPI_GATE_RES, x, gate=g_a, gate_rows=1, gate_ld=6 * D)
            ops.norm_modulate(x, hb, M, D, kind=0, eps=1e-6, shift=sh_m, scale=sc_m, mod_rows=1, mod_ld=6 * D)
            ops.gemm(hb, q['fc1_w'], q['fc1_b'], ops.EPI_GELU_ERF, f1)
            ops.gemm(f1, q['fc2_w'], q['fc2_b'], ops.EPI_GATE_RES, x, gate=g_m, gate_rows=1, gate_ld=6 * D)
        return x


def DiT2_B_2(**kw):
    return DiT2(depth=12, hidden_size=768, patch_size=2, num_heads=12, **kw)


def DiT2_L_2(**kw):
    return DiT2(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kw)


def DiT2_XL_2(**kw):
    return DiT2(depth=28, hidden_size=1152, patch_size=2, num_heads=16, **kw)


DiT2_models = {'DiT2-B/2': DiT2_B_2, 'DiT2-L/2': DiT2_L_2, 'DiT2-XL/2': DiT2_XL_2}
