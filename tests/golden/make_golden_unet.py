"""Golden section for the U-Net denoiser of the ShapeNet / FFHQ entry point (see make_golden.py): the reference's
guided_diffusion.unet.UNetModel (with ldm.modules.attention_compat.SpatialTransformer) and its DDIM loop with v-prediction + LSGM
mixed prediction (`mixing_normal=True`), run from /root/reference in the build container; oracle/unet.py is checked in the same pass.

    python tests/golden/make_golden.py unet
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

from ln3diff_amd.synth import synth_state_dict, synth_input  # noqa: E402
from oracle import unet as ounet, samplers as osamp  # noqa: E402

from unet_configs import CONFIGS, synth_unet_sd  # noqa: E402


def build_ref(cfg):
    from guided_diffusion.unet import UNetModel
    with contextlib.redirect_stdout(io.StringIO()):
        m = UNetModel(image_size=cfg['image_size'], in_channels=cfg['in_channels'], model_channels=cfg['model_channels'],
                      out_channels=cfg['out_channels'], num_res_blocks=cfg['num_res_blocks'],
                      attention_resolutions=tuple(cfg['attention_resolutions']), channel_mult=cfg['channel_mult'], num_heads=cfg['num_heads'],
                      use_scale_shift_norm=cfg['use_scale_shift_norm'], mixed_prediction=True,
                      use_spatial_transformer=cfg['use_spatial_transformer'], transformer_depth=cfg.get('transformer_depth', 1),
                      context_dim=cfg['context_dim'] if cfg['use_spatial_transformer'] else -1, roll_out=cfg['roll_out'])
    return m.eval()


def sec_unet():
    import make_golden as mg
    mg = sys.modules.get('__main__') if hasattr(sys.modules.get('__main__'), 'check') else mg
    check, save, manifest_json = mg.check, mg.save, mg.manifest_json
    print('== U-Net denoiser (reference guided_diffusion.unet.UNetModel vs oracle.unet)')
    models = {}
    for tag, cfg in CONFIGS.items():
        m = build_ref(cfg)
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = synth_unet_sd(shapes, 0)
        m.load_state_dict(sd, strict=True)
        B = 1 if tag == 'shapenet' else 2
        C = cfg['in_channels'] * (3 if cfg['roll_out'] else 1)
        x = synth_input('x', (B, C, cfg['image_size'], cfg['image_size']), 3)
        t = torch.tensor([0.037, 0.911][:B])                       # _WrappedModel hands t / 1000 to the network (respace.py:131)
        ctx = synth_input('c', (B, 77, cfg['context_dim']), 3) if cfg['use_spatial_transformer'] else None
        y = m(x, t, context=ctx)
        yo = ounet.unet_forward(sd, cfg, x, t, ctx)
        check(f'{tag} forward ({len(shapes)} tensors, {sum(int(np.prod(s)) for s in shapes.values()) / 1e6:.1f} M parameters)', yo, y, 1e-5)
        save(f'unet_{tag}', y=y, t=t, manifest=manifest_json(shapes))
        models[tag] = (m, sd, cfg)

    # DDIM with CFG over the tiny transformer U-Net the way the ShapeNet entry point samples: v-prediction, mixing_normal=True,
    # context through apply_model_inference (nsr/lsgm/crossattn_cldm.py:206-211), timestep_respacing ddim25 / eta 0.3
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion.respace import SpacedDiffusion, space_timesteps
    m, sd, cfg = models['tiny_st']
    B = 2
    z = synth_input('z', (B, 4, 16, 16), 41)
    c = synth_input('c', (B, 77, 768), 41)

    class Trainer:                                       # what SpacedDiffusion's _WrappedModel calls (respace.py:124-136)
        def __init__(self):
            self.ddp_model = m

        def apply_model_inference(self, x, t, c, **kw):
            return m(x, t, context=c['c_crossattn'])
    for spec, eta, s, seed in (('ddim25', 0.3, 3.0, 7), ('ddim10', 0.0, 1.0, 9)):
        diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, spec), betas=gd.get_named_beta_schedule('linear', 1000),
                               model_mean_type=gd.ModelMeanType.V, model_var_type=gd.ModelVarType.FIXED_LARGE,
                               loss_type=gd.LossType.MSE, rescale_timesteps=False)
        torch.manual_seed(seed)
        y_ref = diff.ddim_sample_loop(Trainer(), (B, 4, 16, 16), cond={'c_crossattn': c}, noise=z.clone(), clip_denoised=False, device='cpu',
                                      eta=eta, unconditional_guidance_scale=s, unconditional_conditioning=torch.zeros(1, 77, 768),
                                      mixing_normal=True)
        torch.manual_seed(seed)
        noises = [torch.randn(B, 4, 16, 16) for _ in range(diff.num_timesteps)]
        tab = osamp.SpacedTables(spec)

        def to_eps(v, xin, tin):
            ab = torch.tensor(tab.alphas_cumprod, dtype=torch.float32)[tin].view(-1, 1, 1, 1)
            eps = torch.sqrt(ab) * v + torch.sqrt(1 - ab) * xin
            return ounet.mixed_prediction(eps, xin, sd['mixing_logit'], torch.sqrt(1 - ab))
        y_or = osamp.ddim_sample_loop(lambda x, t, cc: ounet.unet_forward(sd, cfg, x, t, cc), z.clone(), c, tab, eta, s, None, noises,
                                      to_eps=to_eps)
        check(f'U-Net ddim {spec} eta={eta} cfg={s} v-prediction + mixing', y_or, y_ref, 2e-4)
        save(f'unet_ddim_tiny_{spec}', final=y_ref, eta=np.array(eta), scale=np.array(s), noise_seed=np.array(seed))
