"""Text conditioner of the T23D path on the HIP kernels.

Mirrors sgm.modules.encoders.modules.FrozenCLIPEmbedder (/root/reference/sgm/modules/encoders/modules.py:347-405; config
sgm/configs/txt2img-clipl-compat.yaml: layer="last", always_return_pooled=True): `transformer` is the CLIP-L text tower
("openai/clip-vit-large-patch14": 12 layers, width 768, 12 heads, quick-GELU, causal attention, 77 positions) whose
state-dict keys are the hub checkpoint's (`transformer.text_model.*`), so released weights load without renaming.
forward(text | token ids) -> (last_hidden_state [B,77,768] f32, pooler_output [B,768] f32) = cond['crossattn'], cond['vector'].

The arithmetic runs on the same kernels as the DiT: LayerNorm(+affine) -> fused QKV GEMM with head-split epilogue ->
attention kernel with the causal flag -> out-proj GEMM with residual epilogue -> LayerNorm -> fc1 GEMM + quick-GELU
epilogue -> fc2 GEMM with residual epilogue; final LayerNorm in fp32.  The tokenizer is third-party data (BPE vocabulary +
merges files): `forward(text)` uses transformers.CLIPTokenizer when those files are available locally, otherwise pass ids.
"""
import torch
import torch.nn as nn

from .. import ops, _cache
from ..dit.dit_models_xformers import Workspace, bf16, f32


class _Attn(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(D, D) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, D, I):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(D, I), nn.Linear(I, D)


class _Layer(nn.Module):
    def __init__(self, D, I):
        super().__init__()
        self.self_attn = _Attn(D)
        self.layer_norm1 = nn.LayerNorm(D)
        self.mlp = _MLP(D, I)
        self.layer_norm2 = nn.LayerNorm(D)


class _Embeddings(nn.Module):
    def __init__(self, vocab, T, D):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, D)
        self.position_embedding = nn.Embedding(T, D)


class _Encoder(nn.Module):
    def __init__(self, D, I, n):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(D, I) for _ in range(n)])


class _TextModel(nn.Module):
    def __init__(self, vocab, T, D, I, n):
        super().__init__()
        self.embeddings = _Embeddings(vocab, T, D)
        self.encoder = _Encoder(D, I, n)
        self.final_layer_norm = nn.LayerNorm(D)


class CLIPTextModel(nn.Module):
    """Parameter container with the hub checkpoint's key layout (text_model.*)."""

    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                 num_attention_heads=12, max_position_embeddings=77, layer_norm_eps=1e-5, eos_token_id=2):
        super().__init__()
        self.heads, self.eps, self.eos_token_id = num_attention_heads, layer_norm_eps, eos_token_id
        self.text_model = _TextModel(vocab_size, max_position_embeddings, hidden_size, intermediate_size, num_hidden_layers)


class FrozenCLIPEmbedder(nn.Module):
    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, always_return_pooled=False, tokenizer_dir=None, **model_kwargs):
        super().__init__()
        assert layer in self.LAYERS
        if layer == "hidden":
            raise NotImplementedError("layer='hidden' is not used by the released configs")
        self.version, self.device, self.max_length = version, device, max_length
        self.layer, self.return_pooled = layer, always_return_pooled
        self.transformer = CLIPTextModel(**model_kwargs)
        self._tok = None
        self.tokenizer_dir = tokenizer_dir          # directory with vocab.json + merges.txt -> ln3diff_amd.sgm.tokenizer (no hub cache needed)
        self._packed = None
        _cache.watch(self)
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    # ------------------------------------------------------------------ tokenizer (third-party data files)
    def tokenize(self, text):
        if self._tok is None and self.tokenizer_dir:
            from .tokenizer import CLIPTokenizer as OwnTokenizer
            own = OwnTokenizer(self.tokenizer_dir, max_length=self.max_length)
            self._tok = lambda text, **kw: {"input_ids": own(text)}
        if self._tok is None:
            try:
                from transformers import CLIPTokenizer
                tok = CLIPTokenizer.from_pretrained(self.version, local_files_only=True)
                need = self.transformer.text_model.embeddings.token_embedding.num_embeddings
                if len(tok) != need:            # recent transformers build an EMPTY tokenizer when the files are missing
                    raise FileNotFoundError(f"tokenizer has {len(tok)} entries, the text tower expects {need}")
                self._tok = tok
            except Exception as e:                                     # no vocabulary on this machine
                raise RuntimeError(f"CLIP BPE vocabulary for '{self.version}' is not available locally ({e}); "
                                   "pass token ids [B, 77] instead of text") from e
        enc = self._tok(text, truncation=True, max_length=self.max_length, return_length=True,
                        return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    # ------------------------------------------------------------------ packing
    def _ensure_packed(self, dev):
        if _cache.fresh(self._packed, dev, 'dev'):
            return
        tm = self.transformer.text_model
        P = {'dev': dev, 'tok': f32(tm.embeddings.token_embedding.weight, dev), 'pos': f32(tm.embeddings.position_embedding.weight, dev),
             'fw': f32(tm.final_layer_norm.weight, dev), 'fb': f32(tm.final_layer_norm.bias, dev), 'layers': []}
        for l in tm.encoder.layers:
            a = l.self_attn
            P['layers'].append({
                'ln1w': f32(l.layer_norm1.weight, dev), 'ln1b': f32(l.layer_norm1.bias, dev),
                'ln2w': f32(l.layer_norm2.weight, dev), 'ln2b': f32(l.layer_norm2.bias, dev),
                'qkv_w': bf16(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0), dev),
                'qkv_b': f32(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0), dev),
                'o_w': bf16(a.out_proj.weight, dev), 'o_b': f32(a.out_proj.bias, dev),
                'fc1_w': bf16(l.mlp.fc1.weight, dev), 'fc1_b': f32(l.mlp.fc1.bias, dev),
                'fc2_w': bf16(l.mlp.fc2.weight, dev), 'fc2_b': f32(l.mlp.fc2.bias, dev)})
        D = P['tok'].shape[1]
        P['zeros'] = torch.zeros(D, device=dev)
        self._packed, self._ws = _cache.stamp(P, self), Workspace(dev)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, text):
        ids = text if torch.is_tensor(text) else self.tokenize(text)
        dev = text.device if (torch.is_tensor(text) and text.is_cuda) else torch.device(self.device)
        if dev.type != 'cuda':
            raise RuntimeError("ln3diff_amd.FrozenCLIPEmbedder runs on the HIP device only (no CPU fallback)")
        self._ensure_packed(dev)
        P, ws, tm = self._packed, self._ws, self.transformer
        B, T = ids.shape
        D = P['tok'].shape[1]
        H = tm.heads
        Dh = D // H
        assert Dh in (64, 128), "attention kernel head sizes"
        M, tpad = B * T, (T + 63) // 64 * 64
        ids_dev = ids.to(device=dev, dtype=torch.int32).contiguous()
        x = ws.get('x', (M, D), torch.float32)
        ops.embed_tokens(ids_dev, P['tok'], P['pos'], x, B, T, D)
        h = ws.get('h', (M, D), torch.bfloat16)
        q = ws.get('q', (B, H, tpad, Dh), torch.bfloat16, zero=True)
        k = ws.get('k', (B, H, tpad, Dh), torch.bfloat16, zero=True)
        vt = ws.get('vt', (B, H, Dh, tpad), torch.bfloat16, zero=True)
        o = ws.get('o', (M, D), torch.bfloat16)
        f1 = ws.get('f1', (M, P['layers'][0]['fc1_w'].shape[0]), torch.bfloat16)
        for L in P['layers']:
            ops.norm_modulate(x, h, M, D, kind=0, eps=tm.eps, weight=L['ln1w'], shift=L['ln1b'], scale=P['zeros'], mod_rows=M, mod_ld=0)
            ops.gemm(h, L['qkv_w'], L['qkv_b'], ops.EPI_HEADS, q, k, vt, M=M, tokens=T, tok_pad=tpad, heads=H, head_dim=Dh,
                     transpose_mask=0b100)
            ops.attention(q, k, vt, o, B, H, T, tpad, T, tpad, Dh, scale=Dh ** -0.5, causal=True)
            ops.gemm(o, L['o_w'], L['o_b'], ops.EPI_GATE_RES, x)
            ops.norm_modulate(x, h, M, D, kind=0, eps=tm.eps, weight=L['ln2w'], shift=L['ln2b'], scale=P['zeros'], mod_rows=M, mod_ld=0)
            ops.gemm(h, L['fc1_w'], L['fc1_b'], ops.EPI_QUICK_GELU, f1)
            ops.gemm(f1, L['fc2_w'], L['fc2_b'], ops.EPI_GATE_RES, x)
        last = torch.empty(B, T, D, device=dev, dtype=torch.float32)
        ops.layernorm_f32(x, P['fw'], P['fb'], last, M, D, tm.eps)
        if tm.eos_token_id == 2:                                        # legacy hub config: position of the largest id
            pos = ids_dev.argmax(-1)
        else:
            pos = (ids_dev == tm.eos_token_id).int().argmax(-1)
        pooled = last[torch.arange(B, device=dev), pos]
        z = last if self.layer == "last" else pooled[:, None, :]
        return (z, pooled) if self.return_pooled else z

    def encode(self, text):
        return self(text)
