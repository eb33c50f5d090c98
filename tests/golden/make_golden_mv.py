"""Fixtures of the multi-view conditioner (FrozenDinov2ImageEmbedderMVPlucker, the released mv23d-plucker configs):
  (1) Pluecker ray maps: the reference's OWN gen_rays / get_plucker_ray (imported from /root/reference through ref_shims and called
      as unbound methods - the class constructor needs torch.hub's dinov2, which is not in the image) on random posed cameras,
      compared with oracle/vit_image.py::plucker_rays, stored sub-sampled;
  (2) the 9-channel DINOv2-reg backbone: transformers' architecture-identical Dinov2WithRegistersModel(num_channels=9) against the
      oracle at a tiny size and at ViT-B/14 (the released arch), on the conditioner's assembled 9-channel input;
  (3) the whole conditioner through the oracle (inputs from (name, shape, seed)), tokens stored sub-sampled.
  python tests/golden/make_golden_mv.py"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_shims                                      # noqa: E402
from make_golden_vit import dino_shapes, dino_to_hf, synth_vit_state_dict, rel, TOK_STRIDE   # noqa: E402
from oracle import vit_image as ovit                  # noqa: E402
from ln3diff_amd.synth import synth_input            # noqa: E402


def cameras(B, T, seed):
    """Random orbit-like posed views: c2w with a rotation about y and x, radius 1.8, intrinsics fx = fy = 1.1, cx = cy = 0.5."""
    g = torch.Generator().manual_seed(seed)
    c = torch.zeros(B, T, 25)
    for b in range(B):
        for t in range(T):
            a, e = float(torch.rand(1, generator=g)) * 6.283, (float(torch.rand(1, generator=g)) - 0.5) * 1.2
            ca, sa, ce, se = np.cos(a), np.sin(a), np.cos(e), np.sin(e)
            Ry = torch.tensor([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], dtype=torch.float32)
            Rx = torch.tensor([[1, 0, 0], [0, ce, -se], [0, se, ce]], dtype=torch.float32)
            R = Ry @ Rx
            m = torch.eye(4)
            m[:3, :3] = R
            m[:3, 3] = R @ torch.tensor([0.0, 0.0, -1.8])
            c[b, t, :16] = m.reshape(-1)
            c[b, t, 16:] = torch.tensor([1.1, 0, 0.5, 0, 1.1 + 0.05 * t, 0.5, 0, 0, 1])
    return c


def run(name, D, n, heads, S, R, B, T):
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    P, G = 14, S // 14
    shapes = dino_shapes(D, n, G, P, R)
    shapes['patch_embed.proj.weight'] = (D, 9, P, P)
    sd = synth_vit_state_dict(shapes, 0)
    img_c = {'img': synth_input('mvimg', (B, T + 1, 3, S, S), 7).clamp(-1, 1), 'c': cameras(B, T + 1, 7)}     # one view more than n_cond_frames
    tok = ovit.dinov2_mv_plucker_forward(sd, img_c, heads, n_cond_frames=T, size=S)
    # the backbone against transformers on the SAME assembled 9-channel input
    x = img_c['img'][:, :T].reshape(B * T, 3, S, S)
    x = ((x + 1) / 2 - torch.tensor((0.485, 0.456, 0.406)).view(1, 3, 1, 1)) / torch.tensor((0.229, 0.224, 0.225)).view(1, 3, 1, 1)
    x = torch.cat([x, ovit.plucker_rays(img_c['c'][:, :T].reshape(B * T, 25), S)], 1)
    cfg = Dinov2WithRegistersConfig(hidden_size=D, num_hidden_layers=n, num_attention_heads=heads, image_size=S, patch_size=P, mlp_ratio=4,
                                    num_register_tokens=R, layer_norm_eps=1e-6, hidden_act='gelu', num_channels=9)
    m = Dinov2WithRegistersModel(cfg).eval()
    m.load_state_dict(dino_to_hf(sd, n), strict=True)
    with torch.no_grad():
        o = m(pixel_values=x).last_hidden_state[:, 1 + R:].reshape(B, T, G * G, D)
    e = rel(tok, o)
    print(f'[{"OK " if e < 2e-5 else "BAD"}] mv-plucker {name}: oracle vs transformers (9-channel Dinov2WithRegisters) patch tokens {e:.3e}')
    assert e < 2e-5
    np.savez_compressed(os.path.join(HERE, f'mv_plucker_{name}.npz'), tokens=tok[:, :, ::TOK_STRIDE].numpy().astype(np.float16 if D > 256 else np.float32),
                        tok_stride=np.array(TOK_STRIDE), heads=np.array(heads), size=np.array(S), n_cond_frames=np.array(T), c=img_c['c'].numpy(),
                        manifest=np.array(json.dumps({k: list(v) for k, v in shapes.items()})))
    print('  wrote', f'mv_plucker_{name}.npz')


def rays_from_reference():
    ref_shims.install()
    for name in ('torchvision.transforms.v2', 'dnnlib', 'dnnlib.util', 'transformers.models.t5'):      # imports of the module that this path never touches
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                ref_shims._mock(name)
    from sgm.modules.encoders import modules as rm
    cls = rm.FrozenDinov2ImageEmbedderMVPlucker
    me = types.SimpleNamespace(reso_encoder=224)
    me.gen_rays = lambda c: cls.gen_rays(me, c)
    c = cameras(2, 3, 11).reshape(6, 25)
    with torch.no_grad():
        ref = cls.get_plucker_ray(me, c.clone()).float()
    mine = ovit.plucker_rays(c, 224)
    e = rel(mine, ref)
    print(f'[{"OK " if e < 1e-6 else "BAD"}] Pluecker rays: oracle vs the reference\'s get_plucker_ray {e:.3e}')
    assert e < 1e-6
    np.savez_compressed(os.path.join(HERE, 'mv_plucker_rays.npz'), c=c.numpy(), rays=ref[:, :, ::7, ::7].numpy(), stride=np.array(7))
    print('  wrote mv_plucker_rays.npz')


if __name__ == '__main__':
    import transformers                                   # before the shims mock torchvision (transformers probes it at import time)
    from transformers import CLIPTextModel, CLIPTokenizer, Dinov2WithRegistersModel   # noqa: F401
    rays_from_reference()
    run('tiny', 128, 2, 2, 56, 4, 2, 2)
    run('vitb14reg', 768, 12, 12, 224, 4, 1, 4)
