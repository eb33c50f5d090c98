"""CPU side of the U-Net denoiser (no GPU): the oracle against the reference's goldens, and the product module's parameter
surface (state-dict keys + shapes) against the reference class's manifest."""
import numpy as np
import pytest
import torch

from conftest import golden, manifest, rel_l2


def _cfg(tag):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from unet_configs import CONFIGS
    return CONFIGS[tag]


def _product(cfg):
    from ln3diff_amd.guided_diffusion.unet import UNetModel
    return UNetModel(image_size=cfg['image_size'], in_channels=cfg['in_channels'], model_channels=cfg['model_channels'],
                     out_channels=cfg['out_channels'], num_res_blocks=cfg['num_res_blocks'], attention_resolutions=tuple(cfg['attention_resolutions']),
                     channel_mult=cfg['channel_mult'], num_heads=cfg['num_heads'], use_scale_shift_norm=cfg['use_scale_shift_norm'],
                     mixed_prediction=True, use_spatial_transformer=cfg['use_spatial_transformer'], transformer_depth=cfg.get('transformer_depth', 1),
                     context_dim=cfg['context_dim'] if cfg['use_spatial_transformer'] else -1, roll_out=cfg['roll_out'])


@pytest.mark.parametrize("tag", ['tiny_st', 'tiny_attn'])
def test_unet_state_dict_surface_matches_the_reference_class(tag):
    g = golden('unet_' + tag)
    want = manifest(g)
    got = {k: tuple(v.shape) for k, v in _product(_cfg(tag)).state_dict().items()}
    assert got == want


def test_create_unet_builds_the_shapenet_configuration():
    """script_util.create_model's U-Net branch with the released ShapeNet flags (sample_shapenet_*_t23d.sh): attention_resolutions
    "4,2,1" on a 32 x 32 latent means downsample rates (8, 16, 32) - transformers at the 4 x 4 level and in the middle block only."""
    from ln3diff_amd.guided_diffusion.unet import create_unet, SpatialTransformer
    g = golden('unet_shapenet')
    with torch.device('meta'):
        m = create_unet(32, 320, 2, attention_resolutions="4,2,1", num_heads=8, use_scale_shift_norm=True, denoise_in_channels=12,
                        denoise_out_channels=12, mixed_prediction=True, use_spatial_transformer=True, transformer_depth=1, context_dim=768)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == manifest(g)
    assert m.attention_resolutions == (8, 16, 32) and m.channel_mult == (1, 2, 4, 4)
    assert sum(isinstance(x, SpatialTransformer) for x in m.modules()) == 6


@pytest.mark.parametrize("tag", ['tiny_st', 'tiny_attn'])
def test_oracle_unet_matches_reference_golden(tag):
    from ln3diff_amd.synth import synth_input
    from oracle import unet as ounet
    from unet_configs import CONFIGS, synth_unet_sd
    cfg = CONFIGS[tag]
    g = golden('unet_' + tag)
    sd = synth_unet_sd(manifest(g), 0)
    C = cfg['in_channels'] * (3 if cfg['roll_out'] else 1)
    x = synth_input('x', (2, C, cfg['image_size'], cfg['image_size']), 3)
    ctx = synth_input('c', (2, 77, cfg['context_dim']), 3) if cfg['use_spatial_transformer'] else None
    y = ounet.unet_forward(sd, cfg, x, torch.from_numpy(g['t']), ctx)
    assert rel_l2(y, g['y']) < 1e-5
