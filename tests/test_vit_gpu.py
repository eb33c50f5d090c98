"""Image conditioners (open_clip ViT-L/14 visual tower, DINOv2 ViT-L/14-reg) on the HIP kernels vs goldens produced by the
architecture-identical `transformers` models in the build container (tests/golden/make_golden_vit.py)."""
import json

import pytest
import torch

from conftest import golden, rel_l2
from ln3diff_amd.synth import synth_input, synth_state_dict, synth_vit_state_dict

pytestmark = pytest.mark.gpu


def _shapes(g):
    return {k: tuple(v) for k, v in json.loads(str(g['manifest'])).items()}


@pytest.mark.parametrize("name,B", [("tiny", 2), ("vitl14", 1)])
def test_openclip_visual_tower_vs_golden(hip_lib, name, B):
    from ln3diff_amd.sgm.image_encoders import FrozenOpenCLIPImageEmbedder
    g = golden(f'vit_clip_{name}')
    sh = _shapes(g)
    D, I = sh['visual.transformer.resblocks.0.mlp.c_fc.weight'][1], sh['visual.transformer.resblocks.0.mlp.c_fc.weight'][0]
    n = 1 + max(int(k.split('.')[3]) for k in sh if '.resblocks.' in k)
    S = int(g['size'])
    m = FrozenOpenCLIPImageEmbedder(output_tokens=True, width=D, mlp_width=I, layers=n, heads=int(g['heads']), image_size=S,
                                    embed_dim=sh['visual.proj'][1], arch="ViT-L-14" if D == 1024 else "custom")
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {'model.' + k: v for k, v in sh.items()}
    m.load_state_dict({'model.' + k: v for k, v in synth_state_dict(sh, 0).items()}, strict=True)
    img = synth_input('img', (B, 3, S, S), 3).cuda()          # the golden fed this tensor to the tower directly
    m.preprocess = lambda x: x
    tokens, pooled = m(img)
    st = int(g['tok_stride'])
    e1 = rel_l2(pooled.cpu(), g['pooled'])
    e2 = rel_l2(tokens[:, ::st].cpu(), torch.from_numpy(g['tokens']).float())
    print('openclip visual', name, e1, e2)
    assert e1 < 2e-2 and e2 < 2e-2, (e1, e2)


@pytest.mark.parametrize("name,B", [("tiny", 2), ("vitl14reg", 1)])
def test_dinov2_tower_vs_golden(hip_lib, name, B):
    from ln3diff_amd.sgm.image_encoders import FrozenDinov2ImageEmbedder
    g = golden(f'vit_dino_{name}')
    sh = _shapes(g)
    D = sh['cls_token'][-1]
    n = 1 + max(int(k.split('.')[1]) for k in sh if k.startswith('blocks.'))
    S = int(g['size'])
    m = FrozenDinov2ImageEmbedder(output_cls=True, width=D, layers=n, heads=int(g['heads']), image_size=S,
                                  num_register_tokens=sh['register_tokens'][1])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {'model.' + k: v for k, v in sh.items()}
    m.load_state_dict({'model.' + k: v for k, v in synth_vit_state_dict(sh, 0).items()}, strict=True)
    img = synth_input('img', (B, 3, S, S), 4).cuda()
    m.preprocess = lambda x: x
    cls, tokens = m(img)
    st = int(g['tok_stride'])
    e1 = rel_l2(cls.cpu(), g['cls'])
    e2 = rel_l2(tokens[:, ::st].cpu(), torch.from_numpy(g['tokens']).float())
    print('dinov2', name, e1, e2)
    assert e1 < 2e-2 and e2 < 2e-2, (e1, e2)


def test_plucker_ray_kernel_vs_reference_fixture(hip_lib):
    """ln3d_plucker_rays against the reference's own get_plucker_ray (tests/golden/mv_plucker_rays.npz, every 7th pixel)."""
    from ln3diff_amd import ops
    g = golden('mv_plucker_rays')
    st = int(g['stride'])
    rays = ops.plucker_rays(torch.from_numpy(g['c']).cuda(), 224)
    assert rays.shape == (6, 6, 224, 224)
    assert rel_l2(rays[:, :, ::st, ::st].cpu(), g['rays']) < 1e-6


@pytest.mark.parametrize("name,B", [("tiny", 2), ("vitb14reg", 1)])
def test_multiview_plucker_conditioner_vs_golden(hip_lib, name, B):
    """r4 (f)3: FrozenDinov2ImageEmbedderMVPlucker (the released mv23d-plucker configs: DINOv2 ViT-B/14-reg, 9-channel patch embedding,
    4 condition views) on the HIP kernels against the transformers-pinned oracle fixture; one view more than n_cond_frames is
    passed (the conditioner takes the first n_cond_frames); reference key layout."""
    from ln3diff_amd.sgm.image_encoders import FrozenDinov2ImageEmbedderMVPlucker, MV23DConditioner
    g = golden(f'mv_plucker_{name}')
    sh = _shapes(g)
    D = sh['cls_token'][-1]
    n = 1 + max(int(k.split('.')[1]) for k in sh if k.startswith('blocks.'))
    S, T = int(g['size']), int(g['n_cond_frames'])
    m = FrozenDinov2ImageEmbedderMVPlucker(arch='vitb', n_cond_frames=T, width=D, layers=n, heads=int(g['heads']), image_size=S,
                                           num_register_tokens=sh['register_tokens'][1])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {'model.' + k: v for k, v in sh.items()}
    assert m.state_dict()['model.patch_embed.proj.weight'].shape[1] == 9
    m.load_state_dict({'model.' + k: v for k, v in synth_vit_state_dict(sh, 0).items()}, strict=True)
    img_c = {'img': synth_input('mvimg', (B, T + 1, 3, S, S), 7).clamp(-1, 1).cuda(), 'c': torch.from_numpy(g['c']).cuda()}
    out = MV23DConditioner(m)(img_c)
    tok = out['concat']
    assert set(out) == {'concat'} and tok.shape == (B, T, (S // 14) ** 2, D)
    e = rel_l2(tok[:, :, ::int(g['tok_stride'])].cpu(), torch.from_numpy(g['tokens']).float())
    print('mv plucker conditioner', name, e)
    assert e < 2e-2, e
    with pytest.raises(ValueError):
        m({'img': img_c['img'][:, :T - 1], 'c': img_c['c'][:, :T - 1]})           # fewer views than n_cond_frames


def test_image_embedder_resizes_other_input_sizes(hip_lib):
    """preprocess (sgm/modules/encoders/modules.py:633-645,802-814) is ONE HIP call (ln3d_image_preprocess): kornia-style Gaussian
    pre-blur when a side shrinks + bicubic / align_corners resize + (x + 1) / 2 + mean / std.  Checked against the CPU restatement
    of kornia's published algorithm (oracle/vit_image.py; kornia itself is absent: unpinned) for shrinking, growing, anisotropic and
    at-size inputs, and through an embedder."""
    from ln3diff_amd import ops
    from ln3diff_amd.sgm.image_encoders import FrozenDinov2ImageEmbedder, FrozenOpenCLIPImageEmbedder
    from oracle import vit_image as ovit
    g = torch.Generator().manual_seed(3)
    mean, std = FrozenOpenCLIPImageEmbedder.MEAN, FrozenOpenCLIPImageEmbedder.STD
    for (H, W, S, aa) in ((64, 64, 56, True), (300, 260, 224, True), (1024, 1024, 224, True), (100, 180, 224, True), (224, 224, 224, True),
                          (300, 260, 224, False), (57, 56, 56, True)):
        x = torch.rand(2, 3, H, W, generator=g) * 2 - 1
        y = ops.image_preprocess(x.cuda(), S, aa, mean, std).cpu()
        y_or = ovit.preprocess(x, S, mean, std, aa)
        e = float((y - y_or).abs().max())
        print('preprocess', (H, W), '->', S, 'antialias', aa, 'max abs diff', e)
        assert y.shape == (2, 3, S, S) and e < 3e-4, e          # fp32 rounding of the source coordinate (scale * index) x image gradient / std
    c = torch.ones(2, 3, 300, 260, device='cuda')                     # a constant image stays constant through blur + bicubic
    want = torch.tensor([(1.0 - m) / s for m, s in zip(mean, std)])
    assert float((ops.image_preprocess(c, 224, True, mean, std).cpu() - want[None, :, None, None]).abs().max()) < 1e-5
    m = FrozenDinov2ImageEmbedder(width=128, layers=1, heads=2, image_size=56)
    x = torch.rand(1, 3, 64, 64, device='cuda') * 2 - 1
    a = m(x)
    ta = a if torch.is_tensor(a) else a[0]
    assert torch.isfinite(ta).all()
    # at-size inputs: the bicubic taps are exactly (0, 1, 0, 0) - the same tokens as feeding the normalised image directly
    x56 = torch.rand(1, 3, 56, 56, device='cuda') * 2 - 1
    pre = m.preprocess(x56)
    assert torch.equal(pre, ops.image_preprocess(x56, 56, False, m.MEAN, m.STD))
    ref = ovit.preprocess(x56.cpu(), 56, m.MEAN, m.STD)
    assert float((pre.cpu() - ref).abs().max()) < 1e-6
