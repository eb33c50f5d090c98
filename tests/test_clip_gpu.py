"""Text conditioner (FrozenCLIPEmbedder = CLIP-L text tower) on the HIP kernels vs goldens produced by the real
`transformers` CLIPTextModel in the build container (tests/golden/make_golden_clip.py)."""
import json

import numpy as np
import pytest
import torch

from conftest import golden, rel_l2
from ln3diff_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu


def _build(g, **kw):
    from ln3diff_amd.sgm.encoders import FrozenCLIPEmbedder
    shapes = {k: tuple(v) for k, v in json.loads(str(g['manifest'])).items()}
    D = shapes['text_model.embeddings.token_embedding.weight'][1]
    vocab = shapes['text_model.embeddings.token_embedding.weight'][0]
    inter = shapes['text_model.encoder.layers.0.mlp.fc1.weight'][0]
    n = 1 + max(int(k.split('.')[3]) for k in shapes if '.layers.' in k)
    m = FrozenCLIPEmbedder(always_return_pooled=True, vocab_size=vocab, hidden_size=D, intermediate_size=inter,
                           num_hidden_layers=n, num_attention_heads=int(g['heads']), eos_token_id=int(g['eos_token_id']), **kw)
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert own == {'transformer.' + k: v for k, v in shapes.items()}, "state-dict keys / shapes must match the hub checkpoint layout"
    m.load_state_dict({'transformer.' + k: v for k, v in synth_state_dict(shapes, 0).items()}, strict=True)
    return m


@pytest.mark.parametrize("name", ["tiny", "tiny_eos", "vitl14"])
def test_clip_text_encoder_vs_transformers_golden(hip_lib, name):
    g = golden(f'clip_text_{name}')
    m = _build(g)
    ids = torch.from_numpy(g['ids'])
    last, pooled = m(ids.cuda())
    e1, e2 = rel_l2(last.cpu(), g['last']), rel_l2(pooled.cpu(), g['pooled'])
    print('clip text', name, e1, e2)
    assert e1 < 2e-2 and e2 < 2e-2, (e1, e2)       # bf16 GEMM operands, fp32 accumulate / residual / norms
    assert last.dtype == torch.float32 and tuple(last.shape) == tuple(g['last'].shape)


def test_clip_text_needs_vocabulary_for_strings(hip_lib):
    m = _build(golden('clip_text_tiny'))
    with pytest.raises(RuntimeError):               # no BPE vocabulary files in this image: loud failure, not a silent default
        m(["a chair"])
