#!/usr/bin/env python
"""Golden vectors for the image conditioners (open_clip ViT-L/14 visual tower, DINOv2 ViT-L/14 with registers).  Both packages
are third-party and not installed; the architecture-identical HuggingFace models ARE, so this script instantiates them with
weights synthesised from (name, shape, seed) under the ORIGINAL packages' key layouts (mapped onto the HF modules), checks
oracle/vit_image.py against them and writes tests/golden/vit_*.npz (outputs + key/shape manifests).
Usage: python tests/golden/make_golden_vit.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ln3diff_amd.synth import synth_input, synth_state_dict, synth_vit_state_dict   # noqa: E402
from oracle import vit_image as ovit                          # noqa: E402


TOK_STRIDE = 4          # fixtures keep every 4th patch token (fp16 at ViT-L size): small files, same coverage of the tower


def rel(a, b):
    return float((a - b).norm() / b.norm())


def clip_shapes(D, I, n, G, P, proj):
    s = {'visual.class_embedding': (D,), 'visual.positional_embedding': (G * G + 1, D), 'visual.proj': (D, proj),
         'visual.conv1.weight': (D, 3, P, P), 'visual.ln_pre.weight': (D,), 'visual.ln_pre.bias': (D,),
         'visual.ln_post.weight': (D,), 'visual.ln_post.bias': (D,)}
    for i in range(n):
        L = f'visual.transformer.resblocks.{i}.'
        s.update({L + 'ln_1.weight': (D,), L + 'ln_1.bias': (D,), L + 'attn.in_proj_weight': (3 * D, D), L + 'attn.in_proj_bias': (3 * D,),
                  L + 'attn.out_proj.weight': (D, D), L + 'attn.out_proj.bias': (D,), L + 'ln_2.weight': (D,), L + 'ln_2.bias': (D,),
                  L + 'mlp.c_fc.weight': (I, D), L + 'mlp.c_fc.bias': (I,), L + 'mlp.c_proj.weight': (D, I), L + 'mlp.c_proj.bias': (D,)})
    return s


def clip_to_hf(sd, n):
    g = lambda k: sd['visual.' + k]
    D = g('class_embedding').shape[0]
    o = {'embeddings.class_embedding': g('class_embedding'), 'embeddings.patch_embedding.weight': g('conv1.weight'),
         'embeddings.position_embedding.weight': g('positional_embedding'), 'pre_layrnorm.weight': g('ln_pre.weight'),
         'pre_layrnorm.bias': g('ln_pre.bias'), 'post_layernorm.weight': g('ln_post.weight'), 'post_layernorm.bias': g('ln_post.bias')}
    for i in range(n):
        L, H = f'transformer.resblocks.{i}.', f'encoder.layers.{i}.'
        w, b = g(L + 'attn.in_proj_weight'), g(L + 'attn.in_proj_bias')
        for j, nm in enumerate(('q_proj', 'k_proj', 'v_proj')):
            o[H + f'self_attn.{nm}.weight'], o[H + f'self_attn.{nm}.bias'] = w[j * D:(j + 1) * D], b[j * D:(j + 1) * D]
        o[H + 'self_attn.out_proj.weight'], o[H + 'self_attn.out_proj.bias'] = g(L + 'attn.out_proj.weight'), g(L + 'attn.out_proj.bias')
        o[H + 'layer_norm1.weight'], o[H + 'layer_norm1.bias'] = g(L + 'ln_1.weight'), g(L + 'ln_1.bias')
        o[H + 'layer_norm2.weight'], o[H + 'layer_norm2.bias'] = g(L + 'ln_2.weight'), g(L + 'ln_2.bias')
        o[H + 'mlp.fc1.weight'], o[H + 'mlp.fc1.bias'] = g(L + 'mlp.c_fc.weight'), g(L + 'mlp.c_fc.bias')
        o[H + 'mlp.fc2.weight'], o[H + 'mlp.fc2.bias'] = g(L + 'mlp.c_proj.weight'), g(L + 'mlp.c_proj.bias')
    return o


def run_clip(name, D, I, n, heads, S, proj, B):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    P, G = 14, S // 14
    shapes = clip_shapes(D, I, n, G, P, proj)
    sd = synth_state_dict(shapes, 0)
    cfg = CLIPVisionConfig(hidden_size=D, intermediate_size=I, num_hidden_layers=n, num_attention_heads=heads, image_size=S,
                           patch_size=P, projection_dim=proj, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    m = CLIPVisionModelWithProjection(cfg).eval()
    hf = clip_to_hf(sd, n)
    own = m.state_dict()
    pre = 'vision_model.' if any(k.startswith('vision_model.') for k in own) else ''
    load = {pre + k: v for k, v in hf.items()}
    load['visual_projection.weight'] = sd['visual.proj'].t().contiguous()
    if pre + 'embeddings.position_ids' in own:
        load[pre + 'embeddings.position_ids'] = own[pre + 'embeddings.position_ids']
    m.load_state_dict(load, strict=True)
    img = synth_input('img', (B, 3, S, S), 3)
    with torch.no_grad():
        o = m(pixel_values=img)
    pooled, tokens, pre_ln = ovit.openclip_visual_forward(sd, img, heads)
    e1, e2 = rel(pooled, o.image_embeds), rel(pre_ln, o.last_hidden_state)
    print(f'[{"OK " if max(e1, e2) < 2e-5 else "BAD"}] clip {name}: oracle vs transformers pooled {e1:.3e} hidden {e2:.3e}')
    assert max(e1, e2) < 2e-5
    np.savez_compressed(os.path.join(HERE, f'vit_clip_{name}.npz'), pooled=pooled.numpy(), tokens=tokens[:, ::TOK_STRIDE].numpy().astype(np.float16 if D > 256 else np.float32), tok_stride=np.array(TOK_STRIDE),
                        heads=np.array(heads), size=np.array(S), manifest=np.array(json.dumps({k: list(v) for k, v in shapes.items()})))
    print('  wrote', f'vit_clip_{name}.npz')


def dino_shapes(D, n, G, P, R, ratio=4):
    s = {'cls_token': (1, 1, D), 'pos_embed': (1, G * G + 1, D), 'register_tokens': (1, R, D), 'mask_token': (1, D),
         'patch_embed.proj.weight': (D, 3, P, P), 'patch_embed.proj.bias': (D,), 'norm.weight': (D,), 'norm.bias': (D,)}
    for i in range(n):
        L = f'blocks.{i}.'
        s.update({L + 'norm1.weight': (D,), L + 'norm1.bias': (D,), L + 'attn.qkv.weight': (3 * D, D), L + 'attn.qkv.bias': (3 * D,),
                  L + 'attn.proj.weight': (D, D), L + 'attn.proj.bias': (D,), L + 'ls1.gamma': (D,), L + 'norm2.weight': (D,),
                  L + 'norm2.bias': (D,), L + 'mlp.fc1.weight': (ratio * D, D), L + 'mlp.fc1.bias': (ratio * D,),
                  L + 'mlp.fc2.weight': (D, ratio * D), L + 'mlp.fc2.bias': (D,), L + 'ls2.gamma': (D,)})
    return s


def dino_to_hf(sd, n):
    D = sd['cls_token'].shape[-1]
    o = {'embeddings.cls_token': sd['cls_token'], 'embeddings.mask_token': sd['mask_token'], 'embeddings.register_tokens': sd['register_tokens'],
         'embeddings.position_embeddings': sd['pos_embed'], 'embeddings.patch_embeddings.projection.weight': sd['patch_embed.proj.weight'],
         'embeddings.patch_embeddings.projection.bias': sd['patch_embed.proj.bias'], 'layernorm.weight': sd['norm.weight'],
         'layernorm.bias': sd['norm.bias']}
    for i in range(n):
        L, H = f'blocks.{i}.', f'encoder.layer.{i}.'
        w, b = sd[L + 'attn.qkv.weight'], sd[L + 'attn.qkv.bias']
        for j, nm in enumerate(('query', 'key', 'value')):
            o[H + f'attention.attention.{nm}.weight'], o[H + f'attention.attention.{nm}.bias'] = w[j * D:(j + 1) * D], b[j * D:(j + 1) * D]
        o[H + 'attention.output.dense.weight'], o[H + 'attention.output.dense.bias'] = sd[L + 'attn.proj.weight'], sd[L + 'attn.proj.bias']
        o[H + 'layer_scale1.lambda1'], o[H + 'layer_scale2.lambda1'] = sd[L + 'ls1.gamma'], sd[L + 'ls2.gamma']
        for a in ('norm1', 'norm2'):
            o[H + a + '.weight'], o[H + a + '.bias'] = sd[L + a + '.weight'], sd[L + a + '.bias']
        for a in ('fc1', 'fc2'):
            o[H + 'mlp.' + a + '.weight'], o[H + 'mlp.' + a + '.bias'] = sd[L + 'mlp.' + a + '.weight'], sd[L + 'mlp.' + a + '.bias']
    return o


def run_dino(name, D, n, heads, S, R, B):
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    P, G = 14, S // 14
    shapes = dino_shapes(D, n, G, P, R)
    sd = synth_vit_state_dict(shapes, 0)
    cfg = Dinov2WithRegistersConfig(hidden_size=D, num_hidden_layers=n, num_attention_heads=heads, image_size=S, patch_size=P,
                                    mlp_ratio=4, num_register_tokens=R, layer_norm_eps=1e-6, hidden_act='gelu')
    m = Dinov2WithRegistersModel(cfg).eval()
    m.load_state_dict(dino_to_hf(sd, n), strict=True)
    img = synth_input('img', (B, 3, S, S), 4)
    with torch.no_grad():
        o = m(pixel_values=img)
    cls, patch = ovit.dinov2_forward(sd, img, heads)
    e1, e2 = rel(cls, o.last_hidden_state[:, 0]), rel(patch, o.last_hidden_state[:, 1 + R:])
    print(f'[{"OK " if max(e1, e2) < 2e-5 else "BAD"}] dino {name}: oracle vs transformers cls {e1:.3e} patch tokens {e2:.3e}')
    assert max(e1, e2) < 2e-5
    np.savez_compressed(os.path.join(HERE, f'vit_dino_{name}.npz'), cls=cls.numpy(), tokens=patch[:, ::TOK_STRIDE].numpy().astype(np.float16 if D > 256 else np.float32), tok_stride=np.array(TOK_STRIDE), heads=np.array(heads),
                        size=np.array(S), manifest=np.array(json.dumps({k: list(v) for k, v in shapes.items()})))
    print('  wrote', f'vit_dino_{name}.npz')


if __name__ == '__main__':
    run_clip('tiny', 128, 256, 2, 2, 56, 64, 2)
    run_dino('tiny', 128, 2, 2, 56, 4, 2)
    run_clip('vitl14', 1024, 4096, 24, 16, 224, 768, 1)
    run_dino('vitl14reg', 1024, 24, 16, 224, 4, 1)
