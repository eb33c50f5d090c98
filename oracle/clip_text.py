"""CPU restatement (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py) of the text conditioner on the T23D path.

sgm.modules.encoders.modules.FrozenCLIPEmbedder (/root/reference/sgm/modules/encoders/modules.py:347-405) wraps the
third-party HuggingFace `CLIPTextModel` ("openai/clip-vit-large-patch14", layer="last", always_return_pooled=True,
sgm/configs/txt2img-clipl-compat.yaml).  The arithmetic therefore lives in `transformers` (not vendored in the reference);
it IS installed in the build container, so this restatement is pinned against it by tests/golden/make_golden_clip.py
(fixtures clip_text_*.npz).  Published algorithm: token + learned position embeddings; pre-LN transformer with CAUSAL
self-attention (no padding mask is passed by the reference), quick-GELU MLP; final LayerNorm; pooled = final-LN state at the
EOS position (legacy configs with eos_token_id == 2: position of the largest token id).
State-dict keys follow the hub checkpoint: text_model.{embeddings,encoder.layers.N.*,final_layer_norm}.
"""
import torch
import torch.nn.functional as F


def clip_text_forward(sd, ids, heads, eos_token_id=2, eps=1e-5, prefix='text_model.'):
    g = lambda k: sd[prefix + k].float()
    B, T = ids.shape
    h = g('embeddings.token_embedding.weight')[ids] + g('embeddings.position_embedding.weight')[:T][None]
    D = h.shape[-1]
    Dh = D // heads
    n_layers = 1 + max(int(k[len(prefix):].split('.')[2]) for k in sd if k.startswith(prefix + 'encoder.layers.'))
    mask = torch.full((T, T), float('-inf')).triu(1)
    for i in range(n_layers):
        L = f'encoder.layers.{i}.'
        r = h
        x = F.layer_norm(h, (D,), g(L + 'layer_norm1.weight'), g(L + 'layer_norm1.bias'), eps)
        q = F.linear(x, g(L + 'self_attn.q_proj.weight'), g(L + 'self_attn.q_proj.bias')) * Dh ** -0.5
        k = F.linear(x, g(L + 'self_attn.k_proj.weight'), g(L + 'self_attn.k_proj.bias'))
        v = F.linear(x, g(L + 'self_attn.v_proj.weight'), g(L + 'self_attn.v_proj.bias'))
        q, k, v = (t.reshape(B, T, heads, Dh).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) + mask, -1) @ v
        a = a.transpose(1, 2).reshape(B, T, D)
        h = r + F.linear(a, g(L + 'self_attn.out_proj.weight'), g(L + 'self_attn.out_proj.bias'))
        r = h
        x = F.layer_norm(h, (D,), g(L + 'layer_norm2.weight'), g(L + 'layer_norm2.bias'), eps)
        x = F.linear(x, g(L + 'mlp.fc1.weight'), g(L + 'mlp.fc1.bias'))
        x = x * torch.sigmoid(1.702 * x)                                  # quick_gelu
        h = r + F.linear(x, g(L + 'mlp.fc2.weight'), g(L + 'mlp.fc2.bias'))
    last = F.layer_norm(h, (D,), g('final_layer_norm.weight'), g('final_layer_norm.bias'), eps)
    if eos_token_id == 2:
        pos = ids.to(torch.int).argmax(-1)
    else:
        pos = (ids.to(torch.int) == eos_token_id).int().argmax(-1)
    return last, last[torch.arange(B), pos]
