"""DiT_I23D_PixelArt (image -> tri-plane latent flow-matching denoiser) on the HIP kernels.

Surface of the reference's dit/dit_i23d.py:173-291 (`DiT_models['DiT-PixArt-L/2'](input_size, num_classes,
learn_sigma, in_channels, context_dim, roll_out, pooling_ctx_dim)`, `forward(x, timesteps, context)` with the sgm
context dict {'crossattn': [B,256,2048] = CLIP(1024) || DINO(1024), 'vector': [B,768]}, `forward_with_cfg`) and its
state-dict keys.  Blocks are ImageCondDiTBlockPixelArtRMSNorm (dit/dit_models_xformers.py:481-539,604-618):
ONE shared adaLN whose output is added to a per-block learned scale_shift_table, RMSNorm pre-norms, self-attention over
[modulated x (768) ; projected DINO tokens (256)] with per-head RMSNorm on q/k, cross-attention to the RMS-normed CLIP
tokens, erf-GELU MLP; T2IFinalLayer.  Everything that depends only on the conditioning image (cap_embedder token,
DINO projection, normed CLIP tokens, every block's cross K / V^T) is computed once per prompt (prepare_context).
"""
import os

import torch
import torch.nn as nn

from .. import ops, _cache
from .dit_models_xformers import (CaptionEmbedder, ImageCondDiTBlock, ImageCondDiTBlockPixelArtRMSNorm, RMSNormP, T2IFinalLayer, bf16, f32,
                                  self_attention_hip, pad_head_columns, attn_head_pad, attn_out_dim)
from .dit_trilatent import DiT, DiT_TriLatent


class DiT_I23D_PixelArt(DiT_TriLatent):
    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, mixing_logit_init=-3, mixed_prediction=True,
                 context_dim=False, pooling_ctx_dim=768, roll_out=False, vit_blk=ImageCondDiTBlockPixelArtRMSNorm,
                 final_layer_blk=T2IFinalLayer):
        super().__init__(input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_ratio,
                         class_dropout_prob, num_classes, learn_sigma, mixing_logit_init, mixed_prediction, context_dim,
                         roll_out, vit_blk, T2IFinalLayer)
        self.clip_ctx_dim = 1024
        del self.clip_text_proj
        self.dino_proj = CaptionEmbedder(context_dim, hidden_size)
        self.clip_spatial_proj = CaptionEmbedder(1024, hidden_size)     # present in the checkpoint, unused by forward
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))
        self.cap_embedder = nn.Sequential(nn.LayerNorm(pooling_ctx_dim), nn.Linear(pooling_ctx_dim, hidden_size))
        self.attention_y_norm = RMSNormP(1024)
        self.pooling_ctx_dim = pooling_ctx_dim

    def _append_proj(self):
        """CaptionEmbedder whose output tokens are appended to the self-attention sequence."""
        return self.dino_proj

    def _ensure_packed(self, device):
        if _cache.fresh(self._packed, device):
            return
        from .dit_models_xformers import Workspace
        D = self.embed_dim
        P = {'device': device}
        self._pack_embedder(P, device)
        P['t_w0'], P['t_b0'] = bf16(self.t_embedder.mlp[0].weight, device), f32(self.t_embedder.mlp[0].bias, device)
        P['t_w2'], P['t_b2'] = bf16(self.t_embedder.mlp[2].weight, device), f32(self.t_embedder.mlp[2].bias, device)
        if getattr(self, 'adaLN_modulation', None) is not None:          # PixArt: ONE shared adaLN + per-block tables
            P['ada_w'], P['ada_b'] = bf16(self.adaLN_modulation[1].weight, device), f32(self.adaLN_modulation[1].bias, device)
            P['sst'] = f32(torch.stack([b.scale_shift_table.reshape(-1) for b in self.blocks], 0), device)   # [depth, 6D]
        else:                                                            # plain DiT_I23D: every block's own adaLN, one GEMM
            P['ada_w'] = bf16(torch.cat([b.adaLN_modulation[1].weight for b in self.blocks], 0), device)    # [depth*6D, D]
            P['ada_b'] = f32(torch.cat([b.adaLN_modulation[1].bias for b in self.blocks], 0), device)
        if hasattr(self, 'cap_embedder'):
            P['cap_ln_w'], P['cap_ln_b'] = f32(self.cap_embedder[0].weight, device), f32(self.cap_embedder[0].bias, device)
            P['cap_w'], P['cap_b'] = bf16(self.cap_embedder[1].weight, device), f32(self.cap_embedder[1].bias, device)
        if hasattr(self, 'attention_y_norm'):
            P['ynorm_w'] = f32(self.attention_y_norm.weight, device)
        if self._append_proj() is not None:
            dp = self._append_proj().y_proj
            P['d_w1'], P['d_b1'] = bf16(dp.fc1.weight, device), f32(dp.fc1.bias, device)
            P['d_w2'], P['d_b2'] = bf16(dp.fc2.weight, device), f32(dp.fc2.bias, device)
        blks = []
        for b in self.blocks:
            q = {}
            q['n1'], q['n2'] = (f32(b.norm1.weight, device), f32(b.norm2.weight, device)) if hasattr(b, 'norm1') else (None, None)
            q['qkv_w'], q['qkv_b'] = bf16(b.attn.qkv.weight, device), f32(b.attn.qkv.bias, device)
            dh = self.embed_dim // self.num_heads
            padw = lambda w: torch.nn.functional.pad(w.detach().float(), (0, attn_head_pad(dh) - dh))     # zero beyond the true head size
            has_qk = getattr(b.attn, 'qk_norm', False)
            q['qn'], q['kn'] = (f32(padw(b.attn.q_norm.weight), device), f32(padw(b.attn.k_norm.weight), device)) if has_qk else (None, None)
            q['proj_w'], q['proj_b'] = bf16(pad_head_columns(b.attn.proj.weight.detach(), self.num_heads, self.embed_dim // self.num_heads), device), f32(b.attn.proj.bias, device)
            q['cq_w'] = bf16(b.cross_attn.to_q.weight, device)
            q['ckv_w'] = bf16(torch.cat([b.cross_attn.to_k.weight, b.cross_attn.to_v.weight], 0), device)
            has_cqk = getattr(b.cross_attn, 'qk_norm', False)
            q['cqn'], q['ckn'] = (f32(b.cross_attn.q_norm.weight, device), f32(b.cross_attn.k_norm.weight, device)) if has_cqk else (None, None)
            if hasattr(b, 'attention_y_norm'):
                q['ynorm'] = f32(b.attention_y_norm.weight, device)
            q['co_w'], q['co_b'] = bf16(b.cross_attn.to_out[0].weight, device), f32(b.cross_attn.to_out[0].bias, device)
            q['fc1_w'], q['fc1_b'] = bf16(b.mlp.mlp[0].weight, device), f32(b.mlp.mlp[1].bias, device)
            q['fc2_w'], q['fc2_b'] = bf16(b.mlp.mlp[2].weight, device), f32(b.mlp.mlp[3].bias, device)
            blks.append(q)
        P['blocks'] = blks
        P['fin_w'], P['fin_b'] = f32(self.final_layer.linear.weight, device), f32(self.final_layer.linear.bias, device)
        P['fin_sst'] = f32(self.final_layer.scale_shift_table, device)          # [2, D]
        P['zeros'] = torch.zeros(max(D, self.pooling_ctx_dim), device=device)
        self._packed = _cache.stamp(P, self)
        self._ws = Workspace(device)

    def _pack_embedder(self, P, device):
        D = self.embed_dim
        P['pe_w'] = f32(self.x_embedder.proj.weight.reshape(D, -1), device)
        P['pe_b'] = f32(self.x_embedder.proj.bias, device)
        P['pos'] = f32(self.pos_embed[0], device)

    def _cls_token(self, vec):
        """pooled token: LayerNorm(affine, eps 1e-5) -> Linear   (dit_i23d.py:211-217,245)"""
        P, ws, D = self._packed, self._ws, self.embed_dim
        Bn = vec.shape[0]
        vn = ws.get('cap_vn', (Bn, self.pooling_ctx_dim), torch.bfloat16)
        ops.norm_modulate(vec.contiguous().float(), vn, Bn, self.pooling_ctx_dim, kind=0, eps=1e-5, weight=P['cap_ln_w'],
                          shift=P['cap_ln_b'], scale=P['zeros'], mod_rows=1 << 30, mod_ld=0)
        cls = torch.empty(Bn, D, device=vec.device, dtype=torch.float32)
        ops.gemm(vn, P['cap_w'], P['cap_b'], ops.EPI_F32, cls)
        return cls

    def _appended_tokens(self, feats):
        """CaptionEmbedder (Linear -> tanh-GELU -> Linear) of [Bn, L, C] features -> bf16 [Bn, L, D]."""
        P, ws, D = self._packed, self._ws, self.embed_dim
        Bn, L, C = feats.shape
        fin = ws.get('app_in', (Bn * L, C), torch.bfloat16)
        ops.cast_bf16(feats.contiguous().float(), fin)
        h1 = ws.get('app_h', (Bn * L, D), torch.bfloat16)
        ops.gemm(fin, P['d_w1'], P['d_b1'], ops.EPI_GELU_TANH, h1)
        out = torch.empty(Bn, L, D, device=feats.device, dtype=torch.bfloat16)
        ops.gemm(h1, P['d_w2'], P['d_b2'], ops.EPI_BF16, out)
        return out

    def _cross_kv(self, ctx_bf16, Bn, Lk):
        """every block's cross-attention K (k_norm applied) / V^T of a [Bn*Lk, C] bf16 context."""
        P, H = self._packed, self.num_heads
        lpad = (Lk + 63) // 64 * 64
        dev = ctx_bf16.device
        k_all = torch.zeros(self.depth, Bn, H, lpad, 64, dtype=torch.bfloat16, device=dev)
        vt_all = torch.zeros(self.depth, Bn, H, 64, lpad, dtype=torch.bfloat16, device=dev)
        for i, q in enumerate(P['blocks']):
            ops.gemm(ctx_bf16, q['ckv_w'], None, ops.EPI_HEADS, k_all[i], vt_all[i], M=Bn * Lk, tokens=Lk, tok_pad=lpad,
                     heads=H, head_dim=64, transpose_mask=0b10)
            if q['ckn'] is not None:
                ops.rmsnorm_heads(k_all[i], q['ckn'], Bn * H * lpad, 64)
        return k_all, vt_all, lpad

    def _fold_uc(self, cc, rows):
        """Samples whose cross-attention context rows are all IDENTICAL - the zero image / text embeddings of the unconditional CFG
        branch (pipeline._zero_uc; every K / V row of a sample is a row-wise function of its context row, so they are identical
        too): softmax over identical scores is uniform whatever the query and the cross-attention sub-block is the constant
        to_out(v) + b per (layer, sample).  For a TRAILING run of such samples (the flow-matching engine's [c, uc] order) the
        constants are computed here, once per prompt, with the same kernels (bf16 V row -> to_out GEMM, fp32 accumulate); forward()
        adds them in the gate / residual epilogue of the self-attention projection and runs to_q / attention / to_out on the leading
        samples only (LN3D_NO_UC_FOLD=1: off).  `rows`: [Bn, L, C] raw context the K / V were made from."""
        cc['fold'] = 0
        Bn = cc['Bn']
        if os.environ.get('LN3D_NO_UC_FOLD') or cc['Lc'] < 1 or Bn < 2:
            return cc
        same = (rows == rows[:, :1]).flatten(1).all(1).tolist()              # one host read per prompt
        fold = 0
        while fold < Bn and same[Bn - 1 - fold]:
            fold += 1
        if not 0 < fold < Bn:
            return cc
        P, H, D = self._packed, self.num_heads, self.embed_dim
        const = torch.zeros(self.depth, Bn, D, dtype=torch.float32, device=rows.device)       # rows < Bn - fold stay 0
        for i, q in enumerate(P['blocks']):
            v_row = cc['vt'][i, Bn - fold:, :, :, 0].reshape(fold, H * 64).contiguous()        # V^T[b, h, d, key 0] = the attention output
            ops.gemm(v_row, q['co_w'], q['co_b'], ops.EPI_F32, const[i, Bn - fold:])
        cc['fold'], cc['const'] = fold, const
        return cc

    @torch.no_grad()
    def prepare_context(self, context):
        ca, vec = context['crossattn'], context['vector']
        dev = ca.device
        self._ensure_packed(dev)
        P, ws = self._packed, self._ws
        Bn, Lc, _ = ca.shape
        D, H, C1 = self.embed_dim, self.num_heads, self.clip_ctx_dim
        want = C1 + self.dino_proj.y_proj.fc1.in_features
        if ca.shape[-1] != want or vec.shape[-1] != self.pooling_ctx_dim:
            raise ValueError(f"context['crossattn'] must be [B, L, {want}] (CLIP {C1} || DINO) and context['vector'] [B, {self.pooling_ctx_dim}]; "
                             f"got {tuple(ca.shape)} / {tuple(vec.shape)}")
        cls = self._cls_token(vec)
        # CLIP tokens: RMSNorm once (dit_i23d.py:247); DINO tokens: tanh-GELU MLP
        clip_n = ws.get('clip_n', (Bn * Lc, C1), torch.bfloat16)
        ops.norm_modulate(ca[..., :C1].contiguous().float(), clip_n, Bn * Lc, C1, kind=1, eps=1e-5, weight=P['ynorm_w'])
        dino_in = ws.get('dino_in', (Bn * Lc, ca.shape[-1] - C1), torch.bfloat16)
        ops.cast_bf16(ca[..., C1:].contiguous().float(), dino_in)
        d1 = ws.get('dino_h', (Bn * Lc, D), torch.bfloat16)
        ops.gemm(dino_in, P['d_w1'], P['d_b1'], ops.EPI_GELU_TANH, d1)
        dino = torch.empty(Bn, Lc, D, device=dev, dtype=torch.bfloat16)
        ops.gemm(d1, P['d_w2'], P['d_b2'], ops.EPI_BF16, dino)
        lpad = (Lc + 63) // 64 * 64
        k_all = torch.zeros(self.depth, Bn, H, lpad, 64, dtype=torch.bfloat16, device=dev)
        vt_all = torch.zeros(self.depth, Bn, H, 64, lpad, dtype=torch.bfloat16, device=dev)
        for i, q in enumerate(P['blocks']):
            ops.gemm(clip_n, q['ckv_w'], None, ops.EPI_HEADS, k_all[i], vt_all[i], M=Bn * Lc, tokens=Lc, tok_pad=lpad,
                     heads=H, head_dim=64, transpose_mask=0b10)
            ops.rmsnorm_heads(k_all[i], q['ckn'], Bn * H * lpad, 64)
        return self._fold_uc({'k': k_all, 'vt': vt_all, 'Lc': Lc, 'lpad': lpad, 'Bn': Bn, 'cls': cls, 'dino': dino}, ca[..., :C1])

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, y=None, get_attr='', context_cache=None, in_scale=None, **kwargs):
        if get_attr != '':
            return getattr(self, get_attr)
        if not x.is_cuda:
            raise RuntimeError("ln3diff_amd.DiT_I23D_PixelArt runs on the HIP device only (no CPU fallback)")
        dev = x.device
        self._ensure_packed(dev)
        P, ws = self._packed, self._ws
        cc = context_cache if context_cache is not None else self.prepare_context(context)
        D, H, depth = self.embed_dim, self.num_heads, self.depth
        Bn, Bx = timesteps.shape[0], x.shape[0]
        assert cc['Bn'] == Bn
        S, p, C = self.input_size, self.patch_size, self.in_channels
        N = self._num_tokens(x)
        Ld = cc['dino'].shape[1]
        NA = N + Ld
        M = Bn * N

        t32 = timesteps.to(device=dev, dtype=torch.float32).contiguous()
        tf = ws.get('tfreq', (Bn, 256), torch.bfloat16)
        ops.timestep_embedding(t32, tf, Bn, 256)
        th = ws.get('th', (Bn, D), torch.bfloat16)
        ops.gemm(tf, P['t_w0'], P['t_b0'], ops.EPI_SILU, th)
        temb = ws.get('temb', (Bn, D), torch.float32)
        ops.gemm(th, P['t_w2'], P['t_b2'], ops.EPI_F32, temb)
        tsum = ws.get('tsum', (Bn, D), torch.float32)                  # t = t_embedder(timesteps) + cap token
        tsilu = ws.get('tsilu', (Bn, D), torch.bfloat16)
        ops.add_act_cast(temb, cc['cls'], tsilu, tsum, Bn * D, 1)
        mod_of, ld = self._modulation(tsilu, Bn)
        nk, neps = self._norm_kind

        xt = self._embed_tokens(x, in_scale, Bx, Bn, N)
        akv = self._appended_kv(cc, Bn, N) if Ld else None             # per-layer K / V^T of the appended tokens, once per prompt
        if akv is None:
            ha = ws.get('ha', (Bn, NA, D), torch.bfloat16)
            if getattr(self, '_ha_src', None) is not cc['dino'] or getattr(self, '_ha_buf', None) is not ha:
                ha[:, N:].copy_(cc['dino'])                            # appended DINO tokens: constant per prompt, and the
                self._ha_src, self._ha_buf = cc['dino'], ha            # norm kernel only ever writes rows < N of each sample
        hb = ws.get('h', (M, D), torch.bfloat16)
        xb = ws.get('xb', (M, D), torch.bfloat16)
        qc = ws.get('qc', (Bn, H, N, 64), torch.bfloat16)
        oc = ws.get('oc', (M, H * 64), torch.bfloat16)
        f1 = ws.get('f1', (M, P['blocks'][0]['fc1_w'].shape[0]), torch.bfloat16)
        probe = getattr(self, '_fc1_probe', None)
        fold = cc.get('fold', 0)                                       # trailing samples with a constant cross-attention output
        Bc = Bn - fold
        Mc = Bc * N
        fuse_cq = ops.heads_norm_fusable(Mc, H * 64, N, 64)
        for i, q in enumerate(P['blocks']):
            mi = mod_of(i)
            if akv is not None:
                # queries / keys / values of the x tokens only (M rows instead of Bn * NA: the appended quarter of the joint
                # sequence never changes); their K / V^T rows land in front of the cached ones of this layer
                sa_k, sa_vt, npad, Dp = akv
                ops.norm_modulate(xt, hb, M, D, kind=nk, eps=neps, weight=q['n1'], shift=mi[:, 0:], scale=mi[:, D:], mod_rows=N, mod_ld=ld)
                qs = ws.get('sa_q', (Bn, H, npad, Dp), torch.bfloat16, zero=True)
                Do = attn_out_dim(D // H)
                ao = ws.get('sa_o', (M, H * Do), torch.bfloat16)
                ops.gemm(hb, q['qkv_w'], q['qkv_b'], ops.EPI_HEADS, qs, sa_k[i], sa_vt[i], M=M, tokens=N, tok_pad=npad, heads=H,
                         head_dim=D // H, transpose_mask=0b100, head_dim_pad=Dp, head_norm0=q['qn'], head_norm1=q['kn'])
                ops.attention(qs, sa_k[i], sa_vt[i], ao, Bn, H, N, npad, NA, npad, Dp, scale=(D // H) ** -0.5,
                              dh_true=(D // H) if Do != Dp else 0)
            else:
                ops.norm_modulate(xt, ha, M, D, kind=nk, eps=neps, weight=q['n1'], shift=mi[:, 0:], scale=mi[:, D:], mod_rows=N,
                                  mod_ld=ld, rows_in=N, rows_out=NA)
                ao = self_attention_hip(ws, 'sa_', ha, Bn, NA, D, H, q['qkv_w'], q['qkv_b'], q['qn'], q['kn'], nq=N)
            ops.gemm(ao, q['proj_w'], q['proj_b'], ops.EPI_GATE_RES, xt, xb, gate=mi[:, 2 * D:], gate_rows=N, gate_ld=ld,
                     res_bias=cc['const'][i] if fold else None, res_bias_ld=D)
            if q['cqn'] is not None and fuse_cq:        # qk_norm of the cross-attention query inside the projection's epilogue
                ops.gemm(xb, q['cq_w'], None, ops.EPI_HEADS, qc, M=Mc, tokens=N, tok_pad=N, heads=H, head_dim=64, head_norm0=q['cqn'])
            else:
                ops.gemm(xb, q['cq_w'], None, ops.EPI_HEADS, qc, M=Mc, tokens=N, tok_pad=N, heads=H, head_dim=64)
                if q['cqn'] is not None:
                    ops.rmsnorm_heads(qc, q['cqn'], Bc * H * N, 64)
            ops.attention(qc, cc['k'][i], cc['vt'][i], oc, Bc, H, N, N, cc['Lc'], cc['lpad'], 64)
            ops.gemm(oc, q['co_w'], q['co_b'], ops.EPI_GATE_RES, xt, M=Mc)
            ops.norm_modulate(xt, hb, M, D, kind=nk, eps=neps, weight=q['n2'], shift=mi[:, 3 * D:], scale=mi[:, 4 * D:],
                              mod_rows=N, mod_ld=ld)
            if probe is not None and i == probe['layer'] and len(probe['events']) < probe['max']:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)      # bench.py measurement hook
                e0.record()
                ops.gemm(hb, q['fc1_w'], q['fc1_b'], ops.EPI_GELU_ERF, f1)
                e1.record()
                probe['events'].append((e0, e1))
            else:
                ops.gemm(hb, q['fc1_w'], q['fc1_b'], ops.EPI_GELU_ERF, f1)
            ops.gemm(f1, q['fc2_w'], q['fc2_b'], ops.EPI_GATE_RES, xt, gate=mi[:, 5 * D:], gate_rows=N, gate_ld=ld)
        return self._output(xt, tsum, Bn, N)

    APPEND_CACHE_MAX_BYTES = 16 << 30

    def _appended_kv(self, cc, Bn, N):
        """The tokens appended to the self-attention sequence (projected DINO / CLIP tokens) are constant per prompt, and so are
        their keys and values in every block: self-attention K / V^T buffers per layer ([depth, Bn, H, NA, Dh], 3.2 GB each at
        network batch 64 - sized for 288 GB of HBM) whose rows [N, NA) are filled ONCE here with the same QKV GEMM + head split
        (+ fused qk-norm) the blocks use; forward() then projects the x tokens only (25 % fewer QKV GEMM rows per layer and step
        at 768 + 256 tokens).  Needs the fused qk-norm epilogue (an unfused norm pass would re-normalise the cached rows every
        step); LN3D_NO_APPEND_CACHE=1 or a cache above APPEND_CACHE_MAX_BYTES fall back to projecting the whole sequence."""
        if 'akv' in cc:
            return cc['akv']
        cc['akv'] = None
        P, ws, D, H = self._packed, self._ws, self.embed_dim, self.num_heads
        Ld = cc['dino'].shape[1]
        Dh = D // H
        Dp = attn_head_pad(Dh)
        NA = N + Ld
        npad = (NA + 63) // 64 * 64
        qk = P['blocks'][0]['qn'] is not None
        if (os.environ.get('LN3D_NO_APPEND_CACHE') or N % 32 or Ld % 32
                or 2 * self.depth * Bn * H * npad * Dp * 2 > self.APPEND_CACHE_MAX_BYTES
                or (qk and not ops.heads_norm_fusable(Bn * N, 3 * H * Dp, N, Dh, Dp))):
            return None
        fused = qk and ops.heads_norm_fusable(Bn * Ld, 3 * H * Dp, Ld, Dh, Dp)    # small prompts batches: norm in a second pass
        dev = cc['dino'].device
        # The prepared context `cc` OWNS these buffers for its lifetime (every live context holds its own copy): the cache is taken
        # only while it fits a third of the device memory that is free right now, and an allocation failure falls back to the
        # full projection as well (several prepared prompts, or a smaller-HBM part, must not die in forward()).
        need = 2 * self.depth * Bn * H * npad * Dp * 2
        try:
            if dev.type == 'cuda' and need > torch.cuda.mem_get_info(dev)[0] // 3:
                return None
            sa_k = torch.zeros(self.depth, Bn, H, npad, Dp, dtype=torch.bfloat16, device=dev)
            sa_vt = torch.zeros(self.depth, Bn, H, Dp, npad, dtype=torch.bfloat16, device=dev)
        except torch.cuda.OutOfMemoryError:
            return None
        qs = ws.get('sa_q', (Bn, H, npad, Dp), torch.bfloat16, zero=True)
        rows = cc['dino'].reshape(Bn * Ld, D)
        for i, q in enumerate(P['blocks']):
            # outputs 1 / 2 start at token N of every (sample, head): the epilogue addresses [b, h, t, :] / [b, h, :, t] with the
            # buffers' own tok_pad stride, so a base pointer moved by N tokens is all it takes
            ops.gemm(rows, q['qkv_w'], q['qkv_b'], ops.EPI_HEADS, qs, sa_k[i][:, :, N:], sa_vt[i][:, :, :, N:], M=Bn * Ld, tokens=Ld,
                     tok_pad=npad, heads=H, head_dim=Dh, transpose_mask=0b100, head_dim_pad=Dp,
                     head_norm0=q['qn'] if fused else None, head_norm1=q['kn'] if fused else None)
            if qk and not fused:       # whole buffer: the x rows are still zero here and stay zero under the norm
                ops.rmsnorm_heads(sa_k[i], q['kn'], Bn * H * npad, Dp, true_dim=Dh)
        qs.zero_()                                                      # the scratch queries of the appended rows are not used
        cc['akv'] = (sa_k, sa_vt, npad, Dp)
        return cc['akv']

    _norm_kind = (1, 1e-5)             # pre-norms: RMSNorm(eps 1e-5) with a weight; the plain DiT_I23D uses affine-free LayerNorm

    def _modulation(self, tsilu, Bn):
        """(i -> [Bn, 6D] view of block i's shift/scale/gate rows, row stride): t2i blocks = shared adaLN(t) + scale_shift_table[i]"""
        P, ws, D, depth = self._packed, self._ws, self.embed_dim, self.depth
        t0 = ws.get('t0', (Bn, 6 * D), torch.float32)
        ops.gemm(tsilu, P['ada_w'], P['ada_b'], ops.EPI_F32, t0)
        mod = ws.get('modb', (depth, Bn, 6 * D), torch.float32)
        ops.add_table_rows(t0, P['sst'], mod, depth, Bn, 6 * D)
        return (lambda i: mod[i]), 6 * D

    # ---- the two ends of forward that the point-cloud variant replaces
    def _num_tokens(self, x):
        return 3 * (self.input_size // self.patch_size) ** 2

    def _embed_tokens(self, x, in_scale, Bx, Bn, N):
        """patchify + shared conv patch-embed + positional embedding -> fp32 tokens [Bn*N, D]"""
        P, D = self._packed, self.embed_dim
        xt = self._ws.get('x', (Bn * N, D), torch.float32)
        ops.patch_embed(x.contiguous().float(), in_scale, P['pe_w'], P['pe_b'], P['pos'], xt, Bx, Bn, self.in_channels, self.input_size,
                        self.patch_size, D)
        return xt

    def _output(self, xt, tsum, Bn, N):
        """T2IFinalLayer (LN, scale_shift_table + t, Linear) + unpatchify -> [Bn, 3*C_out, S, S]"""
        P, D, S = self._packed, self.embed_dim, self.input_size
        out = torch.empty(Bn, self.out_channels * 3, S, S, dtype=torch.float32, device=xt.device)
        ops.final_layer(xt, tsum, tsum, D, P['fin_sst'][0], P['fin_sst'][1], P['fin_w'], P['fin_b'], out, Bn,
                        self.out_channels, S, self.patch_size, D)
        return out

    @torch.no_grad()
    def forward_with_cfg(self, x, t, context=None, cfg_scale=4.0, context_cache=None, **kw):
        eps = self.forward(x, t, context, context_cache=context_cache)
        ops.cfg_combine_dup(eps, float(cfg_scale))
        return eps


class DiT_I23D(DiT_I23D_PixelArt):
    """The plain image-conditioned DiT (reference dit/dit_i23d.py:24-170; registry 'DiT-XL/2', 'DiT-L/2', 'DiT-B/2', 'DiT-B/1' of
    dit_i23d.DiT_models): ImageCondDiTBlock blocks - every block has its OWN adaLN (no shared adaLN / scale_shift_table),
    affine-free LayerNorm(eps 1e-6) pre-norms and its own attention_y_norm for the CLIP tokens; the pooled token is
    clip_text_proj(context['vector']) (CaptionEmbedder, tanh-GELU MLP); T2IFinalLayer.  Runs on the same block machinery as the
    PixArt variant: only the modulation source, the pre-norm kind and the prompt-side preparation differ."""

    _norm_kind = (0, 1e-6)

    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, mixing_logit_init=-3, mixed_prediction=True,
                 context_dim=False, pooling_ctx_dim=768, roll_out=False, vit_blk=ImageCondDiTBlock, final_layer_blk=T2IFinalLayer):
        DiT_TriLatent.__init__(self, input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_ratio, class_dropout_prob,
                               num_classes, learn_sigma, mixing_logit_init, mixed_prediction, context_dim, roll_out, vit_blk,
                               T2IFinalLayer)
        self.clip_ctx_dim = 1024
        self.dino_proj = CaptionEmbedder(context_dim, hidden_size)
        self.clip_spatial_proj = CaptionEmbedder(1024, hidden_size)     # present in the checkpoint, unused by forward
        self.pooling_ctx_dim = self.clip_text_proj.y_proj.fc1.in_features

    def _modulation(self, tsilu, Bn):
        P, ws, D, depth = self._packed, self._ws, self.embed_dim, self.depth
        mod = ws.get('modall', (Bn, depth * 6 * D), torch.float32)
        ops.gemm(tsilu, P['ada_w'], P['ada_b'], ops.EPI_F32, mod)
        return (lambda i: mod[:, i * 6 * D:]), depth * 6 * D

    def _cls_token(self, vec):
        """clip_text_proj(context['vector']): Linear -> tanh-GELU -> Linear, fp32 out (dit_i23d.py:112)"""
        P, ws, D = self._packed, self._ws, self.embed_dim
        Bn = vec.shape[0]
        vin = ws.get('cls_in', (Bn, vec.shape[-1]), torch.bfloat16)
        ops.cast_bf16(vec.contiguous().float(), vin)
        h1 = ws.get('cls_h', (Bn, D), torch.bfloat16)
        ops.gemm(vin, P['c_w1'], P['c_b1'], ops.EPI_GELU_TANH, h1)
        cls = torch.empty(Bn, D, device=vec.device, dtype=torch.float32)
        ops.gemm(h1, P['c_w2'], P['c_b2'], ops.EPI_F32, cls)
        return cls

    def _pack_embedder(self, P, device):
        super()._pack_embedder(P, device)
        cp = self.clip_text_proj.y_proj
        P['c_w1'], P['c_b1'] = bf16(cp.fc1.weight, device), f32(cp.fc1.bias, device)
        P['c_w2'], P['c_b2'] = bf16(cp.fc2.weight, device), f32(cp.fc2.bias, device)

    @torch.no_grad()
    def prepare_context(self, context):
        ca, vec = context['crossattn'], context['vector']
        dev = ca.device
        self._ensure_packed(dev)
        P, ws = self._packed, self._ws
        Bn, Lc, _ = ca.shape
        D, H, C1 = self.embed_dim, self.num_heads, self.clip_ctx_dim
        want = C1 + self.dino_proj.y_proj.fc1.in_features
        if ca.shape[-1] != want or vec.shape[-1] != self.pooling_ctx_dim:
            raise ValueError(f"context['crossattn'] must be [B, L, {want}] (CLIP {C1} || DINO) and context['vector'] [B, {self.pooling_ctx_dim}]; "
                             f"got {tuple(ca.shape)} / {tuple(vec.shape)}")
        cls = self._cls_token(vec)
        dino = self._appended_tokens(ca[..., C1:])
        lpad = (Lc + 63) // 64 * 64
        k_all = torch.zeros(self.depth, Bn, H, lpad, 64, dtype=torch.bfloat16, device=dev)
        vt_all = torch.zeros(self.depth, Bn, H, 64, lpad, dtype=torch.bfloat16, device=dev)
        clip = ca[..., :C1].contiguous().float()
        cn = ws.get('clip_n', (Bn * Lc, C1), torch.bfloat16)
        for i, q in enumerate(P['blocks']):
            ops.norm_modulate(clip, cn, Bn * Lc, C1, kind=1, eps=1e-5, weight=q['ynorm'])            # the BLOCK's attention_y_norm
            ops.gemm(cn, q['ckv_w'], None, ops.EPI_HEADS, k_all[i], vt_all[i], M=Bn * Lc, tokens=Lc, tok_pad=lpad, heads=H, head_dim=64,
                     transpose_mask=0b10)
            ops.rmsnorm_heads(k_all[i], q['ckn'], Bn * H * lpad, 64)
        return self._fold_uc({'k': k_all, 'vt': vt_all, 'Lc': Lc, 'lpad': lpad, 'Bn': Bn, 'cls': cls, 'dino': dino}, ca[..., :C1])


class DiT_I23D_PixelArt_MVCond(DiT_I23D_PixelArt):
    """Multi-view image conditioned variant (reference dit/dit_i23d.py:293-384): the projected CLIP spatial tokens
    (clip_spatial_proj) are appended to the self-attention sequence, the flattened multi-view DINO features context['concat']
    [B, V, L, C] are the cross-attention context (raw: no attention_y_norm, no projection); `dino_proj` does not exist.
    Same blocks / kernels as the single-view model: only the prompt-side preparation differs."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        del self.dino_proj

    def _append_proj(self):
        return self.clip_spatial_proj

    @torch.no_grad()
    def prepare_context(self, context):
        ca, vec, mv = context['crossattn'], context['vector'], context['concat']
        if ca.shape[-1] != self.clip_spatial_proj.y_proj.fc1.in_features or mv.dim() != 4:
            raise ValueError(f"MVCond context: 'crossattn' [B, L, {self.clip_spatial_proj.y_proj.fc1.in_features}] (CLIP spatial tokens), "
                             f"'concat' [B, V, L, C] (multi-view DINO); got {tuple(ca.shape)} / {tuple(mv.shape)}")
        self._ensure_packed(ca.device)
        ws = self._ws
        Bn = ca.shape[0]
        cls = self._cls_token(vec)
        appended = self._appended_tokens(ca)
        Lk = mv.shape[1] * mv.shape[2]
        mvb = ws.get('mv_in', (Bn * Lk, mv.shape[3]), torch.bfloat16)
        ops.cast_bf16(mv.reshape(Bn * Lk, mv.shape[3]).contiguous().float(), mvb)
        k_all, vt_all, lpad = self._cross_kv(mvb, Bn, Lk)
        return self._fold_uc({'k': k_all, 'vt': vt_all, 'Lc': Lk, 'lpad': lpad, 'Bn': Bn, 'cls': cls, 'dino': appended}, mv.reshape(Bn, Lk, -1))


class DiT_I23D_PixelArt_MVCond_noClip(DiT_I23D_PixelArt):
    """Multi-view variant without any CLIP branch (reference dit/dit_i23d.py:387-492; registered there as
    'DiT-PixArt-MV-L/2'): t = t_embedder(timesteps), nothing is appended to the self-attention sequence
    (ImageCondDiTBlockPixelArtNoclip), cross-attention over the flattened multi-view DINO features context['concat'];
    dino_proj, clip_spatial_proj and cap_embedder do not exist."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        del self.dino_proj
        del self.clip_spatial_proj, self.cap_embedder

    def _append_proj(self):
        return None

    @torch.no_grad()
    def prepare_context(self, context):
        mv = context['concat']
        dev = mv.device
        self._ensure_packed(dev)
        Bn, D = mv.shape[0], self.embed_dim
        Lk = mv.shape[1] * mv.shape[2]
        mvb = self._ws.get('mv_in', (Bn * Lk, mv.shape[3]), torch.bfloat16)
        ops.cast_bf16(mv.reshape(Bn * Lk, mv.shape[3]).contiguous().float(), mvb)
        k_all, vt_all, lpad = self._cross_kv(mvb, Bn, Lk)
        return self._fold_uc({'k': k_all, 'vt': vt_all, 'Lc': Lk, 'lpad': lpad, 'Bn': Bn, 'cls': torch.zeros(Bn, D, device=dev),
                              'dino': torch.zeros(Bn, 0, D, device=dev, dtype=torch.bfloat16)}, mv.reshape(Bn, Lk, -1))


class DiT_TriLatent_PixelArt(DiT_I23D_PixelArt):
    """T23D DiT with PixArt-style blocks (reference dit/dit_trilatent.py:146-270; registry 'DiT-PixelArt-L/2', 'DiT-PixelArt-B/2'):
    t = t_embedder + cap_embedder(context['vector']), one shared adaLN + per-block scale_shift_table, PixelArtTextCondDiTBlock
    (RMSNorm pre-norms, no qk-norm, cross-attention over the text tokens normalised per block), T2IFinalLayer, and the
    `forward_with_cfg(x, t, context=..., cfg_scale=...)` the flow-matching engine calls - i.e. the T23D denoiser of that engine.
    Runs on the I23D block machinery: nothing is appended to the self-attention sequence, the per-block normalised context's
    K / V^T are computed once per prompt."""

    def __init__(self, input_size=32, patch_size=2, in_channels=4, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=True, mixing_logit_init=-3, mixed_prediction=True,
                 context_dim=False, roll_out=False, vit_blk=None, final_layer_blk=T2IFinalLayer):
        from .dit_models_xformers import PixelArtTextCondDiTBlock
        DiT_TriLatent.__init__(self, input_size, patch_size, in_channels, hidden_size, depth, num_heads, mlp_ratio, class_dropout_prob,
                               num_classes, learn_sigma, mixing_logit_init, mixed_prediction, context_dim, roll_out,
                               PixelArtTextCondDiTBlock, T2IFinalLayer)
        del self.clip_text_proj
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 6 * hidden_size, bias=True))
        self.cap_embedder = nn.Sequential(nn.LayerNorm(context_dim), nn.Linear(context_dim, hidden_size))
        self.pooling_ctx_dim = context_dim
        self.clip_ctx_dim = context_dim

    def _append_proj(self):
        return None

    @torch.no_grad()
    def prepare_context(self, context):
        ca, vec = context['crossattn'], context['vector']
        if ca.shape[-1] != self.clip_ctx_dim or vec.shape[-1] != self.pooling_ctx_dim:
            raise ValueError(f"context['crossattn'] [B, L, {self.clip_ctx_dim}] text tokens and context['vector'] [B, {self.pooling_ctx_dim}]; "
                             f"got {tuple(ca.shape)} / {tuple(vec.shape)}")
        dev = ca.device
        self._ensure_packed(dev)
        P, ws = self._packed, self._ws
        Bn, Lc, Cd = ca.shape
        D, H = self.embed_dim, self.num_heads
        cls = self._cls_token(vec)
        lpad = (Lc + 63) // 64 * 64
        k_all = torch.zeros(self.depth, Bn, H, lpad, 64, dtype=torch.bfloat16, device=dev)
        vt_all = torch.zeros(self.depth, Bn, H, 64, lpad, dtype=torch.bfloat16, device=dev)
        caf = ca.contiguous().float()
        cn = ws.get('ctx_n', (Bn * Lc, Cd), torch.bfloat16)
        for i, q in enumerate(P['blocks']):
            ops.norm_modulate(caf, cn, Bn * Lc, Cd, kind=1, eps=1e-5, weight=q['ynorm'])       # the BLOCK's attention_y_norm
            ops.gemm(cn, q['ckv_w'], None, ops.EPI_HEADS, k_all[i], vt_all[i], M=Bn * Lc, tokens=Lc, tok_pad=lpad, heads=H, head_dim=64,
                     transpose_mask=0b10)
        return self._fold_uc({'k': k_all, 'vt': vt_all, 'Lc': Lc, 'lpad': lpad, 'Bn': Bn, 'cls': cls,
                              'dino': torch.zeros(Bn, 0, D, device=dev, dtype=torch.bfloat16)}, ca)


def DiT_L_TriLatent_Pixelart_2(**kwargs):         # dit_trilatent.py:311-318 ('DiT-PixelArt-L/2')
    kwargs.pop('vit_blk', None)
    return DiT_TriLatent_PixelArt(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kwargs)


def DiT_B_TriLatent_Pixelart_2(**kwargs):         # dit_trilatent.py:302-309 ('DiT-PixelArt-B/2')
    kwargs.pop('vit_blk', None)
    return DiT_TriLatent_PixelArt(depth=12, hidden_size=768, patch_size=2, num_heads=12, **kwargs)


class DiT_pcd_I23D_PixelArt_MVCond(DiT_I23D_PixelArt_MVCond):
    """Point-cloud latent variant (reference dit/dit_i23d.py:500-588; registry key 'DiT-PixArt-MV-PCD-L'): the tokens are the N
    points of x [B, N, in_channels]; x_embedder is a timm Mlp (Linear -> tanh-GELU -> Linear) instead of the conv patch-embed,
    there is no positional embedding and no unpatchify - the output is [B, N, out_channels].  Conditioning and blocks as MVCond.
    On the HIP path the embedder runs as two GEMMs with the input channels zero-padded to the GEMM's K granule, and the final
    Linear as a bf16 GEMM (the output width is padded to a multiple of 4 and sliced)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        from .dit_models_xformers import Mlp
        self.x_embedder = Mlp(self.in_channels, self.embed_dim, self.embed_dim)
        del self.pos_embed

    def _pack_embedder(self, P, device):
        C, D = self.in_channels, self.embed_dim
        kp = (C + 63) // 64 * 64
        w1 = torch.zeros(D, kp)
        w1[:, :C] = self.x_embedder.fc1.weight.detach().float().cpu()
        P['xe_w1'], P['xe_b1'] = bf16(w1, device), f32(self.x_embedder.fc1.bias, device)
        P['xe_w2'], P['xe_b2'] = bf16(self.x_embedder.fc2.weight, device), f32(self.x_embedder.fc2.bias, device)
        P['xe_kpad'] = kp
        co = self.final_layer.linear.weight.shape[0]
        cp = (co + 3) // 4 * 4
        fw = torch.zeros(cp, D)
        fw[:co] = self.final_layer.linear.weight.detach().float().cpu()
        fb = torch.zeros(cp)
        fb[:co] = self.final_layer.linear.bias.detach().float().cpu()
        P['finp_w'], P['finp_b'], P['finp_n'] = bf16(fw, device), f32(fb, device), co

    def _num_tokens(self, x):
        return x.shape[1]

    def _embed_tokens(self, x, in_scale, Bx, Bn, N):
        P, ws, D = self._packed, self._ws, self.embed_dim
        assert x.dim() == 3 and x.shape[2] == self.in_channels, "point-cloud latent [B, N, in_channels]"
        xin = x.float()
        if in_scale is not None:                       # c_in of a denoiser wrapper, one factor per network row
            xin = xin.repeat(Bn // Bx, 1, 1) * in_scale.view(-1, 1, 1)
        elif Bn != Bx:
            xin = xin.repeat(Bn // Bx, 1, 1)
        xb = ws.get('pcd_in', (Bn * N, P['xe_kpad']), torch.bfloat16, zero=True)
        xb[:, :self.in_channels] = xin.reshape(Bn * N, self.in_channels).to(torch.bfloat16)
        h1 = ws.get('pcd_h', (Bn * N, D), torch.bfloat16)
        ops.gemm(xb, P['xe_w1'], P['xe_b1'], ops.EPI_GELU_TANH, h1)
        xt = ws.get('x', (Bn * N, D), torch.float32)
        ops.gemm(h1, P['xe_w2'], P['xe_b2'], ops.EPI_F32, xt)
        return xt

    def _output(self, xt, tsum, Bn, N):
        P, ws, D = self._packed, self._ws, self.embed_dim
        yb = ws.get('pcd_fin', (Bn * N, D), torch.bfloat16)
        ops.norm_modulate(xt, yb, Bn * N, D, kind=0, eps=1e-6, shift=tsum, scale=tsum, mod_rows=N, mod_ld=D,
                          shift_table=P['fin_sst'][0], scale_table=P['fin_sst'][1])
        out = torch.empty(Bn * N, P['finp_w'].shape[0], device=xt.device, dtype=torch.float32)
        ops.gemm(yb, P['finp_w'], P['finp_b'], ops.EPI_F32, out)
        return out[:, :P['finp_n']].reshape(Bn, N, P['finp_n']).contiguous()


def DiT_L_Pixelart_MV_pcd(**kwargs):          # reference dit_i23d.py:675-681
    return DiT_pcd_I23D_PixelArt_MVCond(depth=24, hidden_size=1024, patch_size=1, num_heads=16, **kwargs)


def DiT_L_Pixelart_MV_2_noclip(**kwargs):
    return DiT_I23D_PixelArt_MVCond_noClip(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kwargs)


def DiT_L_Pixelart_MV_2(**kwargs):
    return DiT_I23D_PixelArt_MVCond(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kwargs)


def DiT_XL_Pixelart_MV_2(**kwargs):       # reference dit_i23d.py:659-664; head size 72 runs in zero-padded 128-wide heads
    return DiT_I23D_PixelArt_MVCond(depth=28, hidden_size=1152, patch_size=2, num_heads=16, **kwargs)


def DiT_B_Pixelart_MV_2(**kwargs):
    return DiT_I23D_PixelArt_MVCond(depth=12, hidden_size=768, patch_size=2, num_heads=12, **kwargs)


def DiT_L_Pixelart_2(**kwargs):
    return DiT_I23D_PixelArt(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kwargs)


def DiT_B_Pixelart_2(**kwargs):
    return DiT_I23D_PixelArt(depth=12, hidden_size=768, patch_size=2, num_heads=12, **kwargs)


def DiT_XL_2(**kwargs):                   # reference dit_i23d.py:597-626: the plain DiT_I23D sizes
    return DiT_I23D(depth=28, hidden_size=1152, patch_size=2, num_heads=16, **kwargs)


def DiT_L_2(**kwargs):
    return DiT_I23D(depth=24, hidden_size=1024, patch_size=2, num_heads=16, **kwargs)


def DiT_B_2(**kwargs):
    return DiT_I23D(depth=12, hidden_size=768, patch_size=2, num_heads=12, **kwargs)


def DiT_B_1(**kwargs):
    return DiT_I23D(depth=12, hidden_size=768, patch_size=1, num_heads=12, **kwargs)


MV_NOCLIP_ARCHS = {'DiT-PixArt-MV-L/2'}        # registry keys built on DiT_I23D_PixelArt_MVCond_noClip (no CLIP tokens in the context)
DiT_models = {'DiT-XL/2': DiT_XL_2, 'DiT-L/2': DiT_L_2, 'DiT-B/2': DiT_B_2, 'DiT-B/1': DiT_B_1,
              'DiT-PixArt-L/2': DiT_L_Pixelart_2, 'DiT-PixArt-B/2': DiT_B_Pixelart_2,
              # reference registry (dit_i23d.py:686-696): 'MV-L/2' is the no-CLIP class, 'MV-B/2' the CLIP+DINO one
              'DiT-PixArt-MV-L/2': DiT_L_Pixelart_MV_2_noclip, 'DiT-PixArt-MV-B/2': DiT_B_Pixelart_MV_2,
              'DiT-PixArt-MVCond-L/2': DiT_L_Pixelart_MV_2, 'DiT-PixArt-MV-XL/2': DiT_XL_Pixelart_MV_2,
              'DiT-PixArt-MV-PCD-L': DiT_L_Pixelart_MV_pcd}
