#!/usr/bin/env python
"""Golden vectors for the CLIP-L text conditioner (FrozenCLIPEmbedder = HuggingFace CLIPTextModel, a third-party dependency
of the reference that is installed in the BUILD container only).  Runs the real `transformers` model with weights
synthesised from (name, shape, seed), checks oracle/clip_text.py against it and writes tests/golden/clip_text_*.npz
(outputs + key/shape manifest only).  Usage: python tests/golden/make_golden_clip.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ln3diff_amd.synth import synth_state_dict   # noqa: E402
from oracle import clip_text as oclip            # noqa: E402


def synth_ids(B, T, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab - 3, (B, T), generator=g)
    ids[:, 0] = vocab - 2                                # <|startoftext|>
    for b in range(B):
        n = int(torch.randint(4, T - 1, (1,), generator=g))
        ids[b, n:] = vocab - 1                           # <|endoftext|> (also the pad token of this tokenizer)
    return ids


def run(name, hidden, layers, heads, inter, vocab, B, eos_cfg):
    from transformers import CLIPTextConfig, CLIPTextModel
    cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                         projection_dim=hidden, pad_token_id=1, bos_token_id=vocab - 2, eos_token_id=eos_cfg)
    m = CLIPTextModel(cfg).eval()
    hf_keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    strip = lambda k: k[len('text_model.'):] if k.startswith('text_model.') else k
    shapes = {'text_model.' + strip(k): s for k, s in hf_keys.items()}      # hub-checkpoint naming
    sd = synth_state_dict(shapes, 0)
    has_prefix = any(k.startswith('text_model.') for k in hf_keys)
    m.load_state_dict({(k if has_prefix else strip(k)): v for k, v in sd.items()}, strict=True)
    ids = synth_ids(B, 77, vocab, 7)
    with torch.no_grad():
        o = m(input_ids=ids)
    last_o, pooled_o = oclip.clip_text_forward(sd, ids, heads, eos_cfg)
    e1 = float((last_o - o.last_hidden_state).norm() / o.last_hidden_state.norm())
    e2 = float((pooled_o - o.pooler_output).norm() / o.pooler_output.norm())
    print(f'[{"OK " if max(e1, e2) < 2e-5 else "BAD"}] {name}: oracle vs transformers last {e1:.3e} pooled {e2:.3e}')
    assert max(e1, e2) < 2e-5
    np.savez_compressed(os.path.join(HERE, f'clip_text_{name}.npz'), ids=ids.numpy().astype(np.int32),
                        last=o.last_hidden_state.numpy(), pooled=o.pooler_output.numpy(), heads=np.array(heads),
                        eos_token_id=np.array(eos_cfg), manifest=np.array(json.dumps({k: list(v) for k, v in shapes.items()})))
    print('  wrote', f'clip_text_{name}.npz')


if __name__ == '__main__':
    torch.manual_seed(0)
    run('tiny', 128, 2, 2, 256, 1000, 2, 2)
    run('tiny_eos', 128, 2, 2, 256, 1000, 2, 999)
    run('vitl14', 768, 12, 12, 3072, 49408, 1, 2)
