"""U-Net denoiser of the ShapeNet / FFHQ entry point (scripts/vit_triplane_diffusion_sample.py) on the HIP kernels.

Same constructor / forward surface and state-dict keys as the reference's guided_diffusion/unet.py:427-791 `UNetModel`
(`forward(x, timesteps, context=None, y=None, get_attr='', **kw)` -> float32 [B, C, H, W]; `mixing_logit` when mixed_prediction),
with its building blocks as parameter containers under the reference's names - `ResBlock` (:164-278), `Downsample` / `Upsample`
(:102-161), `AttentionBlock` (:281-336), and `SpatialTransformer` / `BasicTransformerBlock` / `CrossAttention` / `FeedForward` of
ldm/modules/attention_compat.py:161-277.  The arithmetic is a fixed sequence of HIP launches on channel-last activations
[N, H*W, C] (fp32 stream, bf16 GEMM operands, fp32 accumulation / norms / softmax):

  conv 3x3        ln3d_im2col3x3 (nearest-2x upsample fused) / ln3d_im2col3x3_strided (stride 2) -> ln3d_gemm_bf16 (+ bias / + residual)
  ResBlock        ln3d_groupnorm_any (+SiLU) -> conv -> emb Linear (GEMM) -> ln3d_groupnorm_any with `h + emb` or the scale / shift
                  modulation folded in -> conv with the residual epilogue onto skip(x) (1x1 GEMM when the width changes)
  transformer     GroupNorm -> 1x1 GEMM -> per block: LayerNorm (ln3d_norm_modulate) -> fused q|k|v GEMM -> ln3d_attention_small ->
                  to_out GEMM (residual epilogue); the same against the text context; LayerNorm -> GEGLU (GEMM + ln3d_geglu) -> GEMM
  AttentionBlock  GroupNorm -> qkv GEMM (rows re-ordered at packing so heads are contiguous per q / k / v) -> ln3d_attention_small -> proj
Not built (no released configuration uses them): class conditioning (num_classes), resblock_updown, dims != 2, use_fp16,
predict_codebook_ids; they raise.
"""
import math

import torch
import torch.nn as nn

from .. import ops, _cache
from ..dit.dit_models_xformers import Workspace, bf16, f32, pad_head_columns, self_attention_hip

_MFMA_MIN_TOKENS = 256          # self-attention over at least this many tokens goes to the MFMA attention kernels (r6)


def _mfma_head(dh):
    """Head sizes the MFMA route takes: a multiple of 8 (the head-split GEMM epilogue) up to the attention kernels' 128."""
    return dh % 8 == 0 and dh <= 128


def conv_nd(dims, *a, **k):
    assert dims == 2, "the sampling path is 2-D"
    return nn.Conv2d(*a, **k)


def linear(*a, **k):
    return nn.Linear(*a, **k)


def normalization(channels):                     # guided_diffusion/nn.py:93-100: GroupNorm32(32, channels), eps 1e-5
    return nn.GroupNorm(32, channels)


def zero_module(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


class TimestepEmbedSequential(nn.Sequential):    # unet.py:84-99 (container: the forward walks its children by type)
    pass


class Upsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        assert use_conv
        self.channels, self.out_channels = channels, out_channels or channels
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=1)


class Downsample(nn.Module):
    def __init__(self, channels, use_conv, dims=2, out_channels=None):
        super().__init__()
        assert use_conv, "conv_resample=True (the reference's default and every released configuration)"
        self.channels, self.out_channels = channels, out_channels or channels
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=1)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False, dims=2,
                 use_checkpoint=False, up=False, down=False):
        super().__init__()
        if up or down or use_conv:
            raise NotImplementedError("ResBlock(up / down / use_conv): resblock_updown is False in every released configuration")
        self.channels, self.out_channels = channels, out_channels or channels
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        self.skip_connection = nn.Identity() if self.out_channels == channels else conv_nd(dims, channels, self.out_channels, 1)


class AttentionBlock(nn.Module):
    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        if use_new_attention_order:
            raise NotImplementedError("use_new_attention_order: the legacy order is the reference's default")
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.proj_out = zero_module(nn.Conv1d(channels, channels, 1))


class CrossAttention(nn.Module):                 # attention_compat.py:161-202
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim or query_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        assert glu, "BasicTransformerBlock uses gated_ff=True"
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))


class BasicTransformerBlock(nn.Module):          # attention_compat.py:205-225
    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)


class SpatialTransformer(nn.Module):             # attention_compat.py:228-277
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None):
        super().__init__()
        self.in_channels, self.n_heads, self.d_head = in_channels, n_heads, d_head
        inner = n_heads * d_head
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim)
                                                 for _ in range(depth)])
        self.proj_out = zero_module(nn.Conv2d(inner, in_channels, kernel_size=1))


# ----------------------------------------------------------------------------- packing helpers
def _pack_conv3(conv, dev, cin_pad=None):
    w = conv.weight.detach().float().cpu()                    # [Cout, Cin, 3, 3] -> [Cout, (ky, kx, c)] with c padded to cin_pad
    co, ci = w.shape[0], w.shape[1]
    cp = cin_pad or ci
    kpad = (9 * cp + 63) // 64 * 64
    m = torch.zeros(co, 9, cp)
    m[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci)
    full = torch.zeros(co, kpad)
    full[:, :9 * cp] = m.reshape(co, 9 * cp)
    return {'w': bf16(full, dev), 'b': f32(conv.bias, dev), 'kpad': kpad, 'cin': cp, 'cout': co}


def _pack_lin(w, b, dev):
    w2 = w.detach().reshape(w.shape[0], -1)
    return {'w': bf16(w2, dev), 'b': None if b is None else f32(b, dev), 'cout': w2.shape[0], 'cin': w2.shape[1]}


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions, dropout=0,
                 channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False, use_fp16=False,
                 num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, mixed_prediction=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=-1, n_embed=None, legacy=True, mixing_logit_init=-6, roll_out=False, **kwargs):
        super().__init__()
        if num_classes is not None or resblock_updown or use_fp16 or n_embed is not None or dims != 2 or not conv_resample:
            raise NotImplementedError("UNetModel: num_classes / resblock_updown / use_fp16 / n_embed / dims != 2 / conv_resample=False are "
                                      "not used by the released sampling configurations")
        self.roll_out = roll_out
        if context_dim == -1:
            context_dim = None
        # shape limits of the kernels this module launches, checked here with the reason instead of surfacing as LN3D_ERR_BAD_ARG from the
        # first failing launch: GEMM K % 64 (every conv is an im2col GEMM with K = 9 C padded to 64, so C % 8), GroupNorm(32)
        # groups; where a level attends: LayerNorm width % 128 and <= 1536 (ln3d_norm_modulate, checked in attn() below)
        widths = sorted({int(model_channels * m) for m in channel_mult})
        for w in widths:
            if w % 32 or w % 8:
                raise ValueError(f"UNetModel: channel width {w} (model_channels x channel_mult) must be a multiple of 32 (GroupNorm32, 8-channel im2col)")
        if model_channels % 64:
            raise ValueError(f"UNetModel: model_channels {model_channels} must be a multiple of 64 (time-embedding GEMM K)")
        if context_dim not in (None, -1) and context_dim % 64:
            raise ValueError(f"UNetModel: context_dim {context_dim} must be a multiple of 64 (GEMM K)")
        self.use_spatial_transformer, self.context_dim = bool(use_spatial_transformer), context_dim
        if use_spatial_transformer:
            assert context_dim is not None, "use_spatial_transformer needs context_dim"
        if context_dim is not None:
            assert use_spatial_transformer, "context_dim needs use_spatial_transformer"
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        assert num_heads != -1 or num_head_channels != -1
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions, self.channel_mult = num_res_blocks, attention_resolutions, channel_mult
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.dtype = torch.float32
        self.mixed_prediction = mixed_prediction
        if mixed_prediction:
            self.mixing_logit = nn.Parameter(mixing_logit_init * torch.ones(1, in_channels * 3 if roll_out else in_channels, 1, 1))
        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))

        def attn(ch, heads_arg):
            nonlocal num_heads
            if use_spatial_transformer and (ch % 128 or ch > 1536):
                raise ValueError(f"UNetModel: a transformer at width {ch} - the LayerNorm kernel takes multiples of 128 up to 1536")
            if num_head_channels == -1:
                dim_head = ch // num_heads
            else:
                num_heads = ch // num_head_channels
                dim_head = num_head_channels
            if legacy:
                dim_head = ch // num_heads if use_spatial_transformer else num_head_channels
            if use_spatial_transformer:
                return SpatialTransformer(ch, num_heads, dim_head, depth=transformer_depth, context_dim=context_dim)
            return AttentionBlock(ch, num_heads=heads_arg if heads_arg is not None else num_heads, num_head_channels=dim_head)

        res = lambda cin, cout: ResBlock(cin, ted, dropout, out_channels=cout, dims=dims, use_scale_shift_norm=use_scale_shift_norm)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [res(ch, int(mult * model_channels))]
                ch = int(mult * model_channels)
                if ds in attention_resolutions:
                    layers.append(attn(ch, None))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res(ch, ch), attn(ch, None), res(ch, ch))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [res(ch + ich, int(model_channels * mult))]
                ch = int(model_channels * mult)
                if ds in attention_resolutions:
                    layers.append(attn(ch, num_heads_upsample))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        self._packed, self._ws = None, None
        _cache.watch(self)

    # any parameter change invalidates the packed device copies
    def load_state_dict(self, *a, **k):
        _cache.bump()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        _cache.bump()
        return super()._apply(fn, *a, **k)

    # ------------------------------------------------------------------ packing
    def _pack_layer(self, m, dev):
        gn = lambda g: (f32(g.weight, dev), f32(g.bias, dev), float(g.eps))
        if isinstance(m, nn.Conv2d):
            cin = m.weight.shape[1]
            return ('conv', {'c': _pack_conv3(m, dev, cin_pad=(cin + 7) // 8 * 8), 'cin_raw': cin})
        if isinstance(m, ResBlock):
            q = {'n1': gn(m.in_layers[0]), 'c1': _pack_conv3(m.in_layers[2], dev), 'emb': _pack_lin(m.emb_layers[1].weight, m.emb_layers[1].bias, dev),
                 'n2': gn(m.out_layers[0]), 'c2': _pack_conv3(m.out_layers[3], dev), 'ss': m.use_scale_shift_norm}
            if not isinstance(m.skip_connection, nn.Identity):
                q['skip'] = _pack_lin(m.skip_connection.weight, m.skip_connection.bias, dev)
            return ('res', q)
        if isinstance(m, Downsample):
            return ('down', {'c': _pack_conv3(m.op, dev)})
        if isinstance(m, Upsample):
            return ('up', {'c': _pack_conv3(m.conv, dev)})
        if isinstance(m, SpatialTransformer):
            blocks = []
            for b in m.transformer_blocks:
                ln = lambda n: (f32(n.weight - 1.0, dev), f32(n.bias, dev), float(n.eps))           # y = LN(x) (1 + (w - 1)) + b
                blocks.append({
                    'n1': ln(b.norm1), 'n2': ln(b.norm2), 'n3': ln(b.norm3),
                    'qkv1': _pack_lin(torch.cat([b.attn1.to_q.weight, b.attn1.to_k.weight, b.attn1.to_v.weight], 0), None, dev),
                    'o1': _pack_lin(b.attn1.to_out[0].weight, b.attn1.to_out[0].bias, dev),
                    'o1p': _pack_lin(pad_head_columns(b.attn1.to_out[0].weight.detach(), m.n_heads, m.d_head), b.attn1.to_out[0].bias, dev)
                    if _mfma_head(m.d_head) else None,
                    'q2': _pack_lin(b.attn2.to_q.weight, None, dev),
                    'kv2': _pack_lin(torch.cat([b.attn2.to_k.weight, b.attn2.to_v.weight], 0), None, dev),
                    'o2': _pack_lin(b.attn2.to_out[0].weight, b.attn2.to_out[0].bias, dev),
                    'ff1': _pack_lin(b.ff.net[0].proj.weight, b.ff.net[0].proj.bias, dev),
                    'ff2': _pack_lin(b.ff.net[2].weight, b.ff.net[2].bias, dev)})
            return ('transformer', {'n': gn(m.norm), 'pin': _pack_lin(m.proj_in.weight, m.proj_in.bias, dev),
                                    'pout': _pack_lin(m.proj_out.weight, m.proj_out.bias, dev), 'blocks': blocks, 'heads': m.n_heads,
                                    'dh': m.d_head})
        if isinstance(m, AttentionBlock):
            C, nh = m.channels, m.num_heads
            ch = C // nh
            # legacy order: output rows are [head][q | k | v][ch] (unet.py:378-380); re-ordered to [q | k | v][head][ch] so that every
            # head is a contiguous column range of q, k and v
            idx = torch.arange(3 * C).reshape(nh, 3, ch).permute(1, 0, 2).reshape(-1)
            w = m.qkv.weight.detach().reshape(3 * C, C)[idx]
            return ('attention', {'n': gn(m.norm), 'qkv': _pack_lin(w, m.qkv.bias.detach()[idx], dev),
                                  'proj': _pack_lin(m.proj_out.weight, m.proj_out.bias, dev), 'heads': nh, 'dh': ch,
                                  'projp': _pack_lin(pad_head_columns(m.proj_out.weight.detach().reshape(C, C), nh, ch), m.proj_out.bias, dev)
                                  if _mfma_head(ch) else None})
        raise TypeError(type(m))

    def _ensure_packed(self, dev):
        if _cache.fresh(self._packed, dev):
            return
        P = {'device': dev}
        P['t0'] = _pack_lin(self.time_embed[0].weight, self.time_embed[0].bias, dev)
        P['t2'] = _pack_lin(self.time_embed[2].weight, self.time_embed[2].bias, dev)
        P['inp'] = [[self._pack_layer(m, dev) for m in blk] for blk in self.input_blocks]
        P['mid'] = [self._pack_layer(m, dev) for m in self.middle_block]
        P['out'] = [[self._pack_layer(m, dev) for m in blk] for blk in self.output_blocks]
        P['norm_out'] = (f32(self.out[0].weight, dev), f32(self.out[0].bias, dev), float(self.out[0].eps))
        P['conv_out'] = _pack_conv3(self.out[2], dev)
        if self.mixed_prediction:
            P['mix'] = f32(self.mixing_logit.reshape(-1), dev)
        self._packed = _cache.stamp(P, self)
        self._ws = Workspace(dev)

    # ------------------------------------------------------------------ pieces (h: f32 [N*H*W, C] channel-last)
    def _new(self, rows, cols, dtype=torch.float32):
        return torch.empty(rows, cols, device=self._packed['device'], dtype=dtype)

    def _gn(self, h, nw, N, HW, C, swish, add_row=None, mod=None):
        y = self._new(N * HW, C, torch.bfloat16)
        ops.groupnorm_any(h, nw[0], nw[1], y, N, HW, C, 32, nw[2], swish, add_row=add_row,
                          mod_scale=None if mod is None else mod[0], mod_shift=None if mod is None else mod[1])
        return y

    def _conv3(self, a_bf, N, H, W, pc, out, up=1, stride=1, epi=ops.EPI_F32):
        Ho, Wo = (H * up, W * up) if stride == 1 else ((H - 1) // stride + 1, (W - 1) // stride + 1)
        col = self._new(N * Ho * Wo, pc['kpad'], torch.bfloat16)
        if stride == 1:
            ops.im2col3x3(a_bf, col, N, H, W, pc['cin'], up, pc['kpad'])
        else:
            ops.im2col3x3_strided(a_bf, col, N, H, W, pc['cin'], stride, pc['kpad'])
        ops.gemm(col, pc['w'], pc['b'], epi, out)
        return Ho, Wo

    def _bf(self, h):
        y = torch.empty_like(h, dtype=torch.bfloat16)
        ops.cast_bf16(h, y)
        return y

    def _res(self, h, q, N, H, W, emb_silu):
        cin, cout, HW = q['c1']['cin'], q['c1']['cout'], H * W
        a = self._gn(h, q['n1'], N, HW, cin, True)
        t = self._new(N * HW, cout)
        self._conv3(a, N, H, W, q['c1'], t)
        e = self._new(N, q['emb']['cout'])
        ops.gemm(emb_silu, q['emb']['w'], q['emb']['b'], ops.EPI_F32, e)
        if q['ss']:                                   # GN(h) * (1 + scale) + shift, then SiLU (unet.py:267-271)
            a2 = self._gn(t, q['n2'], N, HW, cout, True, mod=(e[:, :cout].contiguous(), e[:, cout:].contiguous()))
        else:                                         # SiLU(GN(h + emb)) (unet.py:272-273)
            a2 = self._gn(t, q['n2'], N, HW, cout, True, add_row=e)
        if 'skip' in q:
            s = self._new(N * HW, cout)
            ops.gemm(self._bf(h), q['skip']['w'], q['skip']['b'], ops.EPI_F32, s)
        else:
            s = h.clone()                             # the block's input may be a saved skip activation: never updated in place
        self._conv3(a2, N, H, W, q['c2'], s, epi=ops.EPI_GATE_RES)
        return s

    def _self_attend_mfma(self, a_bf, qkv, B, N, heads, dh):
        """r6: self-attention over >= 256 tokens on the MFMA attention kernels (csrc/attention.hip) - the fused q|k|v GEMM splits heads in its
        epilogue (q / k [B, H, N, Dp], V^T [B, H, Dp, N], head size zero-padded to 64 / 128: exact, the pad contributes 0 to q.k and meets zero
        columns of the padded output projection) instead of ln3d_attention_small's one-wavefront-per-query scalar loop (1.3 GFLOP per
        attention at the ShapeNet U-Net's 32 x 32 level).  Reference: guided_diffusion/unet.py:281-389, ldm/modules/attention_compat.py:161-277."""
        return self_attention_hip(self._ws, 'u%d_' % dh, a_bf, B, N, heads * dh, heads, qkv['w'], qkv['b'])

    def _attend(self, qv, kv, vv, B, heads, Nq, Nk, dh, ldq, ldkv):
        o = self._new(B * Nq, heads * dh, torch.bfloat16)
        ops.attention_small(qv, kv, vv, o, B, heads, Nq, Nk, dh, ldq, ldkv, ldkv, dh ** -0.5)
        return o

    def _transformer(self, h, q, N, H, W, ctx_bf, Lc):
        HW, C = H * W, h.shape[1]
        heads, dh = q['heads'], q['dh']
        inner = heads * dh
        rows = N * HW
        a = self._gn(h, q['n'], N, HW, C, False)
        tok = self._new(rows, inner)
        ops.gemm(a, q['pin']['w'], q['pin']['b'], ops.EPI_F32, tok)
        for b in q['blocks']:
            def ln(nw):
                y = self._new(rows, inner, torch.bfloat16)
                ops.norm_modulate(tok, y, rows, inner, kind=0, eps=nw[2], shift=nw[1], scale=nw[0], mod_rows=rows, mod_ld=0)
                return y
            if b['o1p'] is not None and HW >= _MFMA_MIN_TOKENS and HW % 32 == 0:
                o = self._self_attend_mfma(ln(b['n1']), b['qkv1'], N, HW, heads, dh)
                ops.gemm(o, b['o1p']['w'], b['o1p']['b'], ops.EPI_GATE_RES, tok)
            else:
                qkv = self._new(rows, 3 * inner, torch.bfloat16)
                ops.gemm(ln(b['n1']), b['qkv1']['w'], None, ops.EPI_BF16, qkv)
                o = self._attend(qkv, qkv[:, inner:], qkv[:, 2 * inner:], N, heads, HW, HW, dh, 3 * inner, 3 * inner)
                ops.gemm(o, b['o1']['w'], b['o1']['b'], ops.EPI_GATE_RES, tok)
            q2 = self._new(rows, inner, torch.bfloat16)
            ops.gemm(ln(b['n2']), b['q2']['w'], None, ops.EPI_BF16, q2)
            if ctx_bf is None:                        # no context: cross-attention defaults to self-attention (attention_compat.py:183)
                raise NotImplementedError("SpatialTransformer without a context")
            kv = self._new(N * Lc, 2 * inner, torch.bfloat16)
            ops.gemm(ctx_bf, b['kv2']['w'], None, ops.EPI_BF16, kv)
            o = self._attend(q2, kv, kv[:, inner:], N, heads, HW, Lc, dh, inner, 2 * inner)
            ops.gemm(o, b['o2']['w'], b['o2']['b'], ops.EPI_GATE_RES, tok)
            g = self._new(rows, b['ff1']['cout'])
            ops.gemm(ln(b['n3']), b['ff1']['w'], b['ff1']['b'], ops.EPI_F32, g)
            gg = self._new(rows, b['ff1']['cout'] // 2, torch.bfloat16)
            ops.geglu(g, gg, rows, b['ff1']['cout'] // 2)
            ops.gemm(gg, b['ff2']['w'], b['ff2']['b'], ops.EPI_GATE_RES, tok)
        s = h.clone()
        ops.gemm(self._bf(tok), q['pout']['w'], q['pout']['b'], ops.EPI_GATE_RES, s)
        return s

    def _attention(self, h, q, N, H, W):
        HW, C = H * W, h.shape[1]
        a = self._gn(h, q['n'], N, HW, C, False)
        s = h.clone()
        if q['projp'] is not None and HW >= _MFMA_MIN_TOKENS and HW % 32 == 0:
            o = self._self_attend_mfma(a, q['qkv'], N, HW, q['heads'], q['dh'])
            ops.gemm(o, q['projp']['w'], q['projp']['b'], ops.EPI_GATE_RES, s)
            return s
        qkv = self._new(N * HW, 3 * C, torch.bfloat16)
        ops.gemm(a, q['qkv']['w'], q['qkv']['b'], ops.EPI_BF16, qkv)
        o = self._attend(qkv, qkv[:, C:], qkv[:, 2 * C:], N, q['heads'], HW, HW, q['dh'], 3 * C, 3 * C)
        ops.gemm(o, q['proj']['w'], q['proj']['b'], ops.EPI_GATE_RES, s)
        return s

    def _run(self, layers, h, N, H, W, emb_silu, ctx_bf, Lc, x_cl=None):
        for kind, q in layers:
            if kind == 'conv':
                out = self._new(N * H * W, q['c']['cout'])
                self._conv3(x_cl, N, H, W, q['c'], out)
                h = out
            elif kind == 'res':
                h = self._res(h, q, N, H, W, emb_silu)
            elif kind == 'transformer':
                h = self._transformer(h, q, N, H, W, ctx_bf, Lc)
            elif kind == 'attention':
                h = self._attention(h, q, N, H, W)
            elif kind == 'down':
                out = self._new(N * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1), q['c']['cout'])
                H, W = self._conv3(self._bf(h), N, H, W, q['c'], out, stride=2)
                h = out
            elif kind == 'up':
                out = self._new(N * 4 * H * W, q['c']['cout'])
                H, W = self._conv3(self._bf(h), N, H, W, q['c'], out, up=2)
                h = out
        return h, H, W

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, get_attr='', **kwargs):
        if isinstance(context, dict):                  # sgm conditioner compat (unet.py:762-763); the guided_diffusion loops' {'c_crossattn': ..}
            context = context['crossattn'] if 'crossattn' in context else context.get('c_crossattn')
        if get_attr != '':
            return getattr(self, get_attr)
        assert y is None, "the model is not class-conditional"
        if self.use_spatial_transformer and context is None:
            raise ValueError("UNetModel(use_spatial_transformer=True) needs a context [B, L, %s] (cross-attention without one is not built; "
                             "the reference's attn2 would fall back to self-attention)" % self.context_dim)
        if context is not None and self.use_spatial_transformer and (context.dim() != 3 or context.shape[-1] != self.context_dim or context.shape[0] != x.shape[0]):
            raise ValueError("context %s: expected [%d, L, %d]" % (tuple(context.shape), x.shape[0], self.context_dim))
        if not x.is_cuda:
            raise RuntimeError("ln3diff_amd.UNetModel runs on the HIP device only (no CPU fallback)")
        dev = x.device
        self._ensure_packed(dev)
        P = self._packed
        B = x.shape[0]
        if self.roll_out:                              # 'b (n c) h w -> b c h (n w)', n = 3
            _, C3, Hh, Ww = x.shape
            x = x.reshape(B, 3, C3 // 3, Hh, Ww).permute(0, 2, 3, 1, 4).reshape(B, C3 // 3, Hh, 3 * Ww)
        x = x.contiguous().float()
        _, Cin, H, W = x.shape
        # time embedding: sincos(model_channels) -> Linear -> SiLU -> Linear; the ResBlocks consume SiLU(emb)
        tf = self._new(B, self.model_channels, torch.bfloat16)
        ops.timestep_embedding(timesteps.to(dev, torch.float32).contiguous(), tf, B, self.model_channels)
        th = self._new(B, P['t0']['cout'], torch.bfloat16)
        ops.gemm(tf, P['t0']['w'], P['t0']['b'], ops.EPI_SILU, th)
        emb = self._new(B, P['t2']['cout'])
        emb_silu = self._new(B, P['t2']['cout'], torch.bfloat16)
        ops.gemm(th, P['t2']['w'], P['t2']['b'], ops.EPI_F32_SILU, emb, emb_silu)
        ctx_bf, Lc = None, 0
        if context is not None:
            Lc = context.shape[1]
            ctx_bf = self._bf(context.to(dev).float().reshape(B * Lc, -1).contiguous())
        cpad = P['inp'][0][0][1]['c']['cin']
        x_cl = self._new(B * H * W, cpad, torch.bfloat16)
        ops.nchw_to_cl_bf16(x, x_cl, B, Cin, H * W, cpad)
        hs = []
        h = None
        for bi, layers in enumerate(P['inp']):
            h, H, W = self._run(layers, h, B, H, W, emb_silu, ctx_bf, Lc, x_cl=x_cl)
            hs.append((h, H, W))
        h, H, W = self._run(P['mid'], h, B, H, W, emb_silu, ctx_bf, Lc)
        for layers in P['out']:
            hp, _, _ = hs.pop()
            h = torch.cat([h, hp], dim=1)              # channel-last: the skip concat is along the last axis
            h, H, W = self._run(layers, h, B, H, W, emb_silu, ctx_bf, Lc)
        a = self._gn(h, P['norm_out'], B, H * W, h.shape[1], True)
        o_cl = self._new(B * H * W, self.out_channels)
        self._conv3(a, B, H, W, P['conv_out'], o_cl)
        out = torch.empty(B, self.out_channels, H, W, device=dev, dtype=torch.float32)
        ops.cl_to_nchw_f32(o_cl, out, B, self.out_channels, H * W)
        if self.roll_out:                              # 'b c h (n w) -> b (n c) h w'
            out = out.reshape(B, self.out_channels, H, 3, W // 3).permute(0, 3, 1, 2, 4).reshape(B, 3 * self.out_channels, H, W // 3).contiguous()
        return out

    @torch.no_grad()
    def mix(self, eps, x, sqrt_one_minus_ab):
        """get_mixed_prediction on this model's eps output, in place (gaussian_diffusion.py:336-348): x = the noisy input x_t."""
        B, C = eps.shape[:2]
        ops.mix_prediction(eps, x.contiguous().float(), self._packed['mix'], float(sqrt_one_minus_ab), B, C, eps.shape[2] * eps.shape[3])
        return eps


def create_unet(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, attention_resolutions="16", num_heads=1,
                num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, denoise_in_channels=-1,
                denoise_out_channels=3, mixed_prediction=False, use_spatial_transformer=False, transformer_depth=1, context_dim=None,
                legacy=True, mixing_logit_init=-6, roll_out=False, **_):
    """guided_diffusion/script_util.py:255-451 `create_model`, the U-Net branch: default channel multipliers per latent size and
    attention_resolutions given as latent sizes ("4,2,1" -> downsample rates image_size // res)."""
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 448: (0.5, 1, 1, 2, 2, 4, 4), 320: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4),
                 224: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4), 32: (1, 2, 4, 4), 16: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    else:
        channel_mult = tuple(int(m) for m in channel_mult.split(","))
    attention_ds = tuple(image_size // int(r) for r in attention_resolutions.split(","))
    return UNetModel(image_size=image_size, in_channels=denoise_in_channels, model_channels=num_channels,
                     out_channels=denoise_out_channels if not learn_sigma else denoise_out_channels * 2, num_res_blocks=num_res_blocks,
                     attention_resolutions=attention_ds, dropout=dropout, channel_mult=channel_mult, num_heads=num_heads,
                     num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample, use_scale_shift_norm=use_scale_shift_norm,
                     mixed_prediction=mixed_prediction, use_spatial_transformer=use_spatial_transformer, transformer_depth=transformer_depth,
                     context_dim=context_dim, legacy=legacy, mixing_logit_init=mixing_logit_init, roll_out=roll_out)
