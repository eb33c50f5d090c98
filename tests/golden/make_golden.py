#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own Python
(imported from /root/reference through ref_shims.py) and, in the same pass, check that
oracle/ reproduces it.  Runs ONLY in the build container (the reference does not travel).

    python tests/golden/make_golden.py [section ...]

Weights and inputs are synthesised deterministically from (name, shape, seed)
(ln3diff_amd/synth.py), so a fixture holds only small outputs plus the shape manifest;
the GPU box regenerates the identical weights/inputs.  What is data here is: inputs'
recipe (names/seeds), expected outputs, and key/shape manifests of the reference
modules.  No reference source text is stored.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402

ref_shims.install()
dmx = ref_shims.ref_dit_modules()

from ln3diff_amd.synth import synth_state_dict, synth_input, orbit_cameras  # noqa: E402
from oracle import dit as odit, samplers as osamp, render as orender, decoder as odec  # noqa: E402

torch.set_grad_enabled(False)
TOL = 2e-5


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, got, want, tol=TOL):
    e = relerr(got, want)
    flag = 'OK ' if e <= tol else 'BAD'
    print(f'  [{flag}] {name}: rel-L2 {e:.3e} (max|d| {float((got - want).abs().max()):.3e})')
    assert e <= tol, name
    return e


def load_synth(module, seed=0):
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    computed = {k: v for k, v in sd.items() if 'pos_embed' in k}
    new = synth_state_dict(shapes, seed, computed)
    module.load_state_dict(new, strict=True)
    return {k: v.clone() for k, v in module.state_dict().items()}, shapes


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)')


def manifest_json(shapes):
    return np.frombuffer(json.dumps({k: list(v) for k, v in shapes.items()}).encode(), dtype=np.uint8)


# ------------------------------------------------------------------ T23D
def build_t23d(hidden, depth, heads):
    from dit.dit_trilatent import DiT_TriLatent
    from dit.dit_models_xformers import TextCondDiTBlock
    m = DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=hidden, depth=depth,
                      num_heads=heads, num_classes=0, learn_sigma=False, context_dim=768,
                      roll_out=True, vit_blk=TextCondDiTBlock)
    return m.eval()


def sec_t23d():
    print('== T23D DiT forward (reference DiT_TriLatent vs oracle.dit.t23d_forward)')
    # tiny
    m = build_t23d(128, 2, 2)
    pe_ref = m.pos_embed.clone()
    check('pos_embed(3,256) D=128', odit.trilatent_pos_embed(128), pe_ref, 1e-6)
    sd, shapes = load_synth(m, 0)
    x = synth_input('x', (2, 12, 32, 32), 0)
    t = torch.tensor([999., 37.])
    ctx = synth_input('ctx', (2, 77, 768), 0)
    y_ref = m(x, t, ctx)
    y_or = odit.t23d_forward(sd, x, t, ctx, 2)
    check('tiny D128 L2 forward', y_or, y_ref)
    save('t23d_tiny', y=y_ref, t=t, manifest=manifest_json(shapes))

    for arch, B in (('DiT-B/2', 1), ('DiT-L/2', 2), ('DiT-XL/2', 1)):
        hidden, depth, heads = odit.DIT_CONFIGS[arch]
        t0 = time.time()
        m = build_t23d(hidden, depth, heads)
        sd, shapes = load_synth(m, 0)
        x = synth_input('x', (B, 12, 32, 32), 0)
        t = torch.tensor([500., 999.][:B])
        ctx = synth_input('ctx', (B, 77, 768), 0)
        y_ref = m(x, t, ctx)
        y_or = odit.t23d_forward(sd, x, t, ctx, heads)
        check(f'{arch} forward B={B}', y_or, y_ref)
        tag = arch.replace('/', '').replace('-', '_').lower()
        save(f't23d_{tag}', y=y_ref, t=t, manifest=manifest_json(shapes))
        print(f'  ({arch}: {time.time() - t0:.1f}s)')
        if arch == 'DiT-B/2':
            sec_config1(m, sd, heads)
        del m, sd


# ------------------------------------------------- config 1: DiT-B/2 50-step p_sample_loop
def sec_config1(m, sd, heads):
    print('== BASELINE config 1: DiT-B/2, SpacedDiffusion("50").p_sample_loop, B=1')
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion.respace import SpacedDiffusion, space_timesteps
    diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, '50'),
                           betas=gd.get_named_beta_schedule('linear', 1000),
                           model_mean_type=gd.ModelMeanType.EPSILON,
                           model_var_type=gd.ModelVarType.FIXED_LARGE,
                           loss_type=gd.LossType.MSE, rescale_timesteps=False)
    tabs = osamp.SpacedTables('50')
    assert tabs.timestep_map == diff.timestep_map
    for nm in ('betas', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod',
               'posterior_mean_coef1', 'posterior_mean_coef2'):
        assert np.array_equal(getattr(tabs, nm), getattr(diff, nm)), nm
    ctx = synth_input('ctx', (1, 77, 768), 1)
    z = synth_input('z', (1, 12, 32, 32), 1)

    class Adapter:                       # what the engines expose to _WrappedModel
        def apply_model_inference(self, x, t, c, **kw):
            return m(x, t, c)
    torch.manual_seed(1234)
    t0 = time.time()
    y_ref = diff.p_sample_loop(Adapter(), (1, 12, 32, 32), cond=ctx, noise=z.clone(),
                               clip_denoised=False, mixing_normal=False, device='cpu')
    print(f'  reference loop {time.time() - t0:.1f}s')
    torch.manual_seed(1234)
    noises = [torch.randn(1, 12, 32, 32) for _ in range(50)]
    trace = []
    y_or = osamp.ddpm_p_sample_loop(lambda x, t, c: odit.t23d_forward(sd, x, t, c, heads),
                                    z.clone(), noises, ctx, tabs, trace=trace)
    check('config1 final latent', y_or, y_ref, 1e-4)
    save('config1_ditb2_ddpm50', final=y_ref, step0=trace[0], step24=trace[24],
         noise_seed=np.array(1234))


# ------------------------------------------------------------- EDM / DDPM on tiny DiT
def sec_samplers():
    print('== samplers on the tiny T23D DiT')
    m = build_t23d(128, 2, 2)
    sd, _ = load_synth(m, 0)
    B = 2
    z = synth_input('z', (B, 12, 32, 32), 41)
    cond = {'crossattn': synth_input('c', (B, 77, 768), 41), 'vector': synth_input('v', (B, 768), 41)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}

    # --- sgm EulerEDMSampler + DiscreteDenoiser(EpsScaling, LegacyDDPM) + VanillaCFG(6.5)
    from sgm.modules.diffusionmodules.sampling import EulerEDMSampler
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    import sgm.util as sgm_util
    dc = {'target': 'sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization'}
    for steps in (250, 10):
        sampler = EulerEDMSampler(discretization_config=dc, num_steps=steps,
                                  guider_config={'target': 'sgm.modules.diffusionmodules.guiders.VanillaCFG',
                                                 'params': {'scale': 6.5}}, device='cpu')
        den = DiscreteDenoiser(scaling_config={'target': 'sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling'},
                               num_idx=1000, discretization_config=dc, do_append_zero=False,
                               quantize_c_noise=True, flip=True)
        sig_ref = sampler.discretization(steps, device='cpu')
        check(f'sigma table n={steps}', osamp.legacy_ddpm_sigmas(steps), sig_ref, 0)
        check('denoiser table', osamp.discrete_denoiser_table(), den.sigmas, 0)
        seen_t = []

        def net(x, t, c, **kw):
            seen_t.append(t.clone())
            return m(x, t, c)
        y_ref = sampler(lambda x, s, c: den(net, x, s, c), z.clone(), cond, uc)
        trace = []
        y_or = osamp.edm_euler_sample(lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2),
                                      z.clone(), cond, uc, steps, 6.5, trace)
        check(f'EulerEDM {steps} steps final latent', y_or, y_ref, 2e-4)
        save(f'edm_tiny_{steps}', final=y_ref, first=trace[0], mid=trace[steps // 2],
             sigmas=sig_ref, idx_first=seen_t[0], idx_last=seen_t[-1])

    # --- r6: the same sampler with noise injection (s_churn > 0: sampling.py:82-130); the reference draws randn_like(x) per churned step
    steps, churn = 10, dict(s_churn=4.0, s_tmin=0.5, s_tmax=10.0, s_noise=1.003)
    sampler = EulerEDMSampler(discretization_config=dc, num_steps=steps,
                              guider_config={'target': 'sgm.modules.diffusionmodules.guiders.VanillaCFG', 'params': {'scale': 6.5}},
                              device='cpu', **churn)
    seen_t = []
    torch.manual_seed(11)
    y_ref = sampler(lambda x, s, c: den(net, x, s, c), z.clone(), cond, uc)
    sig = osamp.legacy_ddpm_sigmas(steps)
    churned = [i for i in range(steps) if churn['s_tmin'] <= float(sig[i]) <= churn['s_tmax']]
    torch.manual_seed(11)
    draws = {i: torch.randn(B, 12, 32, 32) for i in churned}                 # the reference's RNG stream, in step order
    trace = []
    y_or = osamp.edm_euler_sample(lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2), z.clone(), cond, uc, steps, 6.5, trace,
                                  step_noise=lambda i: draws[i], **churn)
    check(f'EulerEDM {steps} steps with s_churn final latent', y_or, y_ref, 2e-4)
    print('   churned steps', churned, 'timesteps seen', [int(t[0]) for t in seen_t])
    save('edm_tiny_10_churn', final=y_ref, mid=trace[steps // 2], churned=np.array(churned), noise_seed=np.array(11),
         idx_seen=np.array([int(t[0]) for t in seen_t]), **{k: np.array(v) for k, v in churn.items()})

    # --- guided_diffusion p_sample_loop '250'
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion.respace import SpacedDiffusion, space_timesteps
    for spec in ('250', 'ddim250'):
        assert osamp.space_timesteps(1000, spec) == space_timesteps(1000, spec)
    diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, '250'),
                           betas=gd.get_named_beta_schedule('linear', 1000),
                           model_mean_type=gd.ModelMeanType.EPSILON,
                           model_var_type=gd.ModelVarType.FIXED_LARGE,
                           loss_type=gd.LossType.MSE, rescale_timesteps=False)

    class Adapter:
        def apply_model_inference(self, x, t, c, **kw):
            return m(x, t, c)
    torch.manual_seed(7)
    y_ref = diff.p_sample_loop(Adapter(), (B, 12, 32, 32), cond=cond['crossattn'], noise=z.clone(),
                               clip_denoised=False, mixing_normal=False, device='cpu')
    torch.manual_seed(7)
    noises = [torch.randn(B, 12, 32, 32) for _ in range(250)]
    tabs = osamp.SpacedTables('250')
    y_or = osamp.ddpm_p_sample_loop(lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2),
                                    z.clone(), noises, cond['crossattn'], tabs)
    check('p_sample_loop 250 final latent', y_or, y_ref, 2e-4)
    save('ddpm_tiny_250', final=y_ref, noise_seed=np.array(7))


# ------------------------------------------------------------------ I23D
def build_i23d(hidden, depth, heads):
    from dit.dit_i23d import DiT_I23D_PixelArt
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        m = DiT_I23D_PixelArt(input_size=32, patch_size=2, in_channels=4, hidden_size=hidden, depth=depth,
                              num_heads=heads, num_classes=0, learn_sigma=False, context_dim=1024,
                              roll_out=True, pooling_ctx_dim=768)
    return m.eval()


def sec_i23d():
    print('== I23D DiT-PixArt forward_with_cfg + flow-matching Euler')
    m = build_i23d(128, 2, 2)
    sd, shapes = load_synth(m, 0)
    B = 2
    x = synth_input('x', (2 * B, 12, 32, 32), 0)
    t = torch.tensor([0.3, 0.3, 0.3, 0.3])
    ctx = {'crossattn': synth_input('ca', (2 * B, 256, 2048), 0), 'vector': synth_input('v', (2 * B, 768), 0)}
    y_ref = m.forward_with_cfg(x, t, ctx, 4.0)
    y_or = odit.i23d_forward_with_cfg(sd, x, t, ctx, 4.0, 2)
    check('tiny I23D forward_with_cfg', y_or, y_ref)
    save('i23d_tiny', y=y_ref, t=t, manifest=manifest_json(shapes))

    # flow matching: transport.Sampler(...).sample_ode('euler', 50)
    from transport import create_transport, Sampler
    tr = create_transport(path_type='Linear', prediction='velocity', snr_type='lognorm')
    z = synth_input('z', (B, 12, 32, 32), 42)
    zs = torch.cat([z, z], 0)
    cond = {'crossattn': synth_input('ca', (B, 256, 2048), 42), 'vector': synth_input('v', (B, 768), 42)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    context = {k: torch.cat([cond[k], uc[k]], 0) for k in cond}      # flow matching: [c, uc]
    for method, steps in (('euler', 50), ('heun', 10)):
        fn = Sampler(tr).sample_ode(sampling_method=method, num_steps=steps)
        y_ref = fn(zs.clone(), m.forward_with_cfg, context=context, cfg_scale=4.0)[-1].chunk(2)[0]
        y_or = osamp.flow_ode_sample(lambda x, t, **kw: odit.i23d_forward_with_cfg(sd, x, t, kw['context'], kw['cfg_scale'], 2),
                                     zs.clone(), steps, method, context=context, cfg_scale=4.0).chunk(2)[0]
        check(f'flow {method} num_steps={steps} final latent', y_or, y_ref, 2e-4)
        save(f'flow_tiny_{method}{steps}', final=y_ref)

    # multi-view conditioned variant (DiT_I23D_PixelArt_MVCond): CLIP spatial tokens appended, flattened MV DINO features cross-attended
    from dit.dit_i23d import DiT_I23D_PixelArt_MVCond
    with torch.no_grad():
        mm = DiT_I23D_PixelArt_MVCond(input_size=32, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2,
                                      num_classes=0, learn_sigma=False, context_dim=768, roll_out=True, pooling_ctx_dim=768).eval()
    sdm, shapes_m = load_synth(mm, 0)
    xm = synth_input('x', (2, 12, 32, 32), 5)
    tm = torch.tensor([0.4, 0.7])
    ctxm = {'crossattn': synth_input('ca', (2, 256, 1024), 5), 'vector': synth_input('v', (2, 768), 5),
            'concat': synth_input('mv', (2, 4, 256, 768), 5)}
    with torch.no_grad():
        ym_ref = mm(xm, tm, ctxm)
    ym_or = odit.i23d_mv_forward(sdm, xm, tm, ctxm, 2)
    check('tiny I23D MVCond forward', ym_or, ym_ref)
    save('i23d_mv_tiny', y=ym_ref, t=tm, manifest=manifest_json(shapes_m))

    from dit.dit_i23d import DiT_I23D_PixelArt_MVCond_noClip          # what the registry calls 'DiT-PixArt-MV-L/2'
    with torch.no_grad():
        mn = DiT_I23D_PixelArt_MVCond_noClip(input_size=32, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2,
                                             num_classes=0, learn_sigma=False, context_dim=768, roll_out=True, pooling_ctx_dim=768).eval()
    sdn, shapes_n = load_synth(mn, 0)
    with torch.no_grad():
        yn_ref = mn(xm, tm, {'concat': ctxm['concat']})
    yn_or = odit.i23d_mv_noclip_forward(sdn, xm, tm, ctxm, 2)
    check('tiny I23D MVCond_noClip forward', yn_or, yn_ref)
    save('i23d_mv_noclip_tiny', y=yn_ref, t=tm, manifest=manifest_json(shapes_n))

    # SDE samplers (transport.Sampler.sample_sde: Euler-Maruyama / Heun; noise from the global CPU generator)
    for method, steps, form, last in (('Euler', 25, 'sigma', 'Mean'), ('Heun', 8, 'linear', 'Euler'), ('Euler', 12, 'decreasing', 'Tweedie')):
        fn = Sampler(tr).sample_sde(sampling_method=method, diffusion_form=form, diffusion_norm=0.7, last_step=last,
                                    last_step_size=0.04, num_steps=steps)
        torch.manual_seed(1234)
        y_ref = fn(zs.clone(), m.forward_with_cfg, context=context, cfg_scale=4.0)[-1].chunk(2)[0]
        torch.manual_seed(1234)
        y_or = osamp.flow_sde_sample(lambda x, t, **kw: odit.i23d_forward_with_cfg(sd, x, t, kw['context'], kw['cfg_scale'], 2),
                                     zs.clone(), steps, method, form, 0.7, last, 0.04, context=context, cfg_scale=4.0)[-1].chunk(2)[0]
        check(f'flow SDE {method} {form} {last} num_steps={steps} final latent', y_or, y_ref, 2e-4)
        save(f'sde_tiny_{method.lower()}{steps}_{form}_{last.lower()}', final=y_ref)

    hidden, depth, heads = odit.DIT_CONFIGS['DiT-L/2']
    m = build_i23d(hidden, depth, heads)
    sd, shapes = load_synth(m, 0)
    x = synth_input('x', (2, 12, 32, 32), 0)
    t = torch.tensor([0.5, 0.5])
    ctx = {'crossattn': synth_input('ca', (2, 256, 2048), 0), 'vector': synth_input('v', (2, 768), 0)}
    y_ref = m.forward_with_cfg(x, t, ctx, 4.0)
    y_or = odit.i23d_forward_with_cfg(sd, x, t, ctx, 4.0, heads)
    check('DiT-PixArt-L/2 forward_with_cfg', y_or, y_ref)
    save('i23d_pixart_l2', y=y_ref, t=t, manifest=manifest_json(shapes))


def sec_ddim():
    print('== DDIM (reference GaussianDiffusion.ddim_sample_loop, CFG on eps) on the tiny T23D DiT')
    m = build_t23d(128, 2, 2)
    sd, _ = load_synth(m, 0)
    from guided_diffusion import gaussian_diffusion as gd
    from guided_diffusion.respace import SpacedDiffusion, space_timesteps
    B = 2
    z = synth_input('z', (B, 12, 32, 32), 41)
    c = synth_input('c', (B, 77, 768), 41)

    class Adapter:
        def apply_model_inference(self, x, t, c, **kw):
            return m(x, t, c['c_crossattn'] if isinstance(c, dict) else c)
    for spec, eta, s, seed in (('ddim50', 0.0, 6.5, 3), ('ddim25', 0.5, 3.0, 5)):
        diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, spec), betas=gd.get_named_beta_schedule('linear', 1000),
                               model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_LARGE,
                               loss_type=gd.LossType.MSE, rescale_timesteps=False)
        torch.manual_seed(seed)
        # the reference repeat_interleaves the null embedding by the batch size: it expects a batch-1 tensor
        y_ref = diff.ddim_sample_loop(Adapter(), (B, 12, 32, 32), cond={'c_crossattn': c}, noise=z.clone(),
                                      clip_denoised=False, device='cpu', eta=eta, unconditional_guidance_scale=s,
                                      unconditional_conditioning=torch.zeros(1, 77, 768))
        torch.manual_seed(seed)
        noises = [torch.randn(B, 12, 32, 32) for _ in range(diff.num_timesteps)]
        y_or = osamp.ddim_sample_loop(lambda x, t, cc: odit.t23d_forward(sd, x, t, cc, 2), z.clone(), c,
                                      osamp.SpacedTables(spec), eta, s, None, noises)
        check(f'ddim {spec} eta={eta} cfg={s}', y_or, y_ref, 2e-4)
        save(f'ddim_tiny_{spec}', final=y_ref, eta=np.array(eta), scale=np.array(s), noise_seed=np.array(seed))


def sec_flow_fixed():
    """The remaining fixed-grid methods transport.integrators.ode hands to odeint (integrators.py:78-119): 'midpoint' and 'rk4' through the
    reference's Sampler.sample_ode on the tiny I23D network.  odeint itself is ref_shims' stand-in (torchdiffeq is absent from the image):
    what is pinned is the reference's grid / drift / CFG plumbing around it; the step formulas restate torchdiffeq 0.2.3 (parity
    unpinned against the package)."""
    print('== flow-matching fixed-grid midpoint / rk4')
    m = build_i23d(128, 2, 2)
    sd, shapes = load_synth(m, 0)
    B = 2
    from transport import create_transport, Sampler
    tr = create_transport(path_type='Linear', prediction='velocity', snr_type='lognorm')
    z = synth_input('z', (B, 12, 32, 32), 42)
    zs = torch.cat([z, z], 0)
    cond = {'crossattn': synth_input('ca', (B, 256, 2048), 42), 'vector': synth_input('v', (B, 768), 42)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    context = {k: torch.cat([cond[k], uc[k]], 0) for k in cond}
    for method, steps in (('midpoint', 10), ('rk4', 6)):
        fn = Sampler(tr).sample_ode(sampling_method=method, num_steps=steps)
        y_ref = fn(zs.clone(), m.forward_with_cfg, context=context, cfg_scale=4.0)[-1].chunk(2)[0]
        y_or = osamp.flow_ode_sample(lambda x, t, **kw: odit.i23d_forward_with_cfg(sd, x, t, kw['context'], kw['cfg_scale'], 2),
                                     zs.clone(), steps, method, context=context, cfg_scale=4.0).chunk(2)[0]
        check(f'flow {method} num_steps={steps} final latent', y_or, y_ref, 2e-4)
        save(f'flow_tiny_{method}{steps}', final=y_ref)


# ------------------------------------------------------------- r6: the other samplers of sgm/modules/diffusionmodules/sampling.py
def sec_more_samplers():
    """HeunEDMSampler (with and without churn), EulerAncestralSampler, DPMPP2SAncestralSampler, DPMPP2MSampler over the tiny T23D DiT:
    the reference's own classes on CPU; the stochastic ones draw randn_like(x) once per step from the global generator (seed stored),
    the oracle twins get the same stream re-drawn."""
    print('== more samplers on the tiny T23D DiT')
    from sgm.modules.diffusionmodules import sampling as S
    from sgm.modules.diffusionmodules.denoiser import DiscreteDenoiser
    m = build_t23d(128, 2, 2)
    sd, _ = load_synth(m, 0)
    B = 2
    z = synth_input('z', (B, 12, 32, 32), 41)
    cond = {'crossattn': synth_input('c', (B, 77, 768), 41), 'vector': synth_input('v', (B, 768), 41)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    dc = {'target': 'sgm.modules.diffusionmodules.discretizer.LegacyDDPMDiscretization'}
    gc = {'target': 'sgm.modules.diffusionmodules.guiders.VanillaCFG', 'params': {'scale': 6.5}}
    den = DiscreteDenoiser(scaling_config={'target': 'sgm.modules.diffusionmodules.denoiser_scaling.EpsScaling'},
                           num_idx=1000, discretization_config=dc, do_append_zero=False, quantize_c_noise=True, flip=True)
    calls = []

    def net(x, t, c, **kw):
        calls.append(int(t[0]))
        return m(x, t, c)
    run = lambda sampler: sampler(lambda x, s, c: den(net, x, s, c), z.clone(), cond, uc)
    onet = lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2)
    steps = 8

    # Heun, deterministic and with churn
    y_ref = run(S.HeunEDMSampler(discretization_config=dc, num_steps=steps, guider_config=gc, device='cpu'))
    n_calls = len(calls)
    tr = []
    y_or = osamp.edm_heun_sample(onet, z.clone(), cond, uc, steps, 6.5, tr)
    check(f'HeunEDM {steps} steps final latent', y_or, y_ref, 2e-4)
    save('heun_tiny_8', final=y_ref, mid=tr[steps // 2], net_calls=np.array(n_calls))
    churn = dict(s_churn=3.0, s_tmin=0.5, s_tmax=10.0, s_noise=1.003)
    torch.manual_seed(12)
    y_ref = run(S.HeunEDMSampler(discretization_config=dc, num_steps=steps, guider_config=gc, device='cpu', **churn))
    sig = osamp.legacy_ddpm_sigmas(steps)
    churned = [i for i in range(steps) if churn['s_tmin'] <= float(sig[i]) <= churn['s_tmax']]
    torch.manual_seed(12)
    draws = {i: torch.randn(B, 12, 32, 32) for i in churned}
    y_or = osamp.edm_heun_sample(onet, z.clone(), cond, uc, steps, 6.5, None, step_noise=lambda i: draws[i], **churn)
    check(f'HeunEDM {steps} steps with s_churn final latent', y_or, y_ref, 2e-4)
    save('heun_tiny_8_churn', final=y_ref, churned=np.array(churned), noise_seed=np.array(12), **{k: np.array(v) for k, v in churn.items()})

    # the ancestral pair: one randn_like(x) per step, in step order
    for name, cls, fn, seed, kw in (('euler_ancestral_tiny_8', S.EulerAncestralSampler, osamp.euler_ancestral_sample, 13, dict(eta=1.0, s_noise=1.0)),
                                    ('euler_ancestral_tiny_8_eta', S.EulerAncestralSampler, osamp.euler_ancestral_sample, 14, dict(eta=0.6, s_noise=1.01)),
                                    ('dpmpp2s_tiny_8', S.DPMPP2SAncestralSampler, osamp.dpmpp2s_ancestral_sample, 15, dict(eta=1.0, s_noise=1.0))):
        torch.manual_seed(seed)
        y_ref = run(cls(discretization_config=dc, num_steps=steps, guider_config=gc, device='cpu', **kw))
        torch.manual_seed(seed)
        draws = [torch.randn(B, 12, 32, 32) for _ in range(steps)]
        tr = []
        y_or = fn(onet, z.clone(), cond, uc, steps, 6.5, step_noise=lambda i: draws[i], trace=tr, **kw)
        check(f'{cls.__name__} {kw} final latent', y_or, y_ref, 2e-4)
        save(name, final=y_ref, mid=tr[steps // 2], noise_seed=np.array(seed), **{k: np.array(v) for k, v in kw.items()})

    # DPM++ 2M (deterministic)
    y_ref = run(S.DPMPP2MSampler(discretization_config=dc, num_steps=steps, guider_config=gc, device='cpu'))
    tr = []
    y_or = osamp.dpmpp2m_sample(onet, z.clone(), cond, uc, steps, 6.5, tr)
    check(f'DPMPP2M {steps} steps final latent', y_or, y_ref, 2e-4)
    save('dpmpp2m_tiny_8', final=y_ref, mid=tr[steps // 2])

    # IdentityGuider (the reference's default without a guider_config: no guidance, the batch is not doubled) under Euler and Heun.  The oracle
    # has one guider: scale 1 with uc = c is the same arithmetic up to the rounding of x_u + (x_c - x_u)
    for name, cls, fn in (('euler_identity_tiny_8', S.EulerEDMSampler, osamp.edm_euler_sample), ('heun_identity_tiny_8', S.HeunEDMSampler, osamp.edm_heun_sample)):
        sampler = cls(discretization_config=dc, num_steps=steps, device='cpu')
        assert type(sampler.guider).__name__ == 'IdentityGuider'
        n0 = len(calls)
        y_ref = run(sampler)
        y_or = fn(onet, z.clone(), cond, cond, steps, 1.0)
        check(f'{cls.__name__} with IdentityGuider final latent', y_or, y_ref, 2e-4)
        save(name, final=y_ref, net_calls=np.array(len(calls) - n0))

    # the other denoiser scalings (denoiser_scaling.py:14-59) and the continuous Denoiser (denoiser.py:13-42), under Euler + CFG
    from sgm.modules.diffusionmodules.denoiser import Denoiser
    sc = 'sgm.modules.diffusionmodules.denoiser_scaling.'
    for name, kind, discrete, qcn in (('euler_vscaling_tiny_8', 'v', True, True), ('euler_vscaling_edmcnoise_tiny_8', 'v_edm', True, False),
                                      ('euler_edmscaling_cont_tiny_8', 'edm', False, False)):
        target = {'v': 'VScaling', 'v_edm': 'VScalingWithEDMcNoise', 'edm': 'EDMScaling'}[kind]
        d2 = (DiscreteDenoiser(scaling_config={'target': sc + target}, num_idx=1000, discretization_config=dc, do_append_zero=False,
                               quantize_c_noise=qcn, flip=True) if discrete else Denoiser(scaling_config={'target': sc + target}))
        labels = []

        def net2(x, t, c, **kw):
            labels.append(float(t[0]))
            return m(x, t, c)
        sampler = S.EulerEDMSampler(discretization_config=dc, num_steps=steps, guider_config=gc, device='cpu')
        y_ref = sampler(lambda x, s, c: d2(net2, x, s, c), z.clone(), cond, uc)
        y_or = osamp.edm_euler_sample(onet, z.clone(), cond, uc, steps, 6.5, scaling=kind, discrete=discrete, quantize_c_noise=qcn)
        check(f'EulerEDM + {target} ({"discrete" if discrete else "continuous"}) final latent', y_or, y_ref, 2e-4)
        save(name, final=y_ref, labels=np.array(labels, dtype=np.float32))

    # EDMDiscretization (discretizer.py:27-39): the product's class against the reference's, bit for bit (no fixture needed)
    from sgm.modules.diffusionmodules.discretizer import EDMDiscretization as RefEDM
    from ln3diff_amd.sgm.sampling import EDMDiscretization as OurEDM
    for n_ in (5, 10, 50):
        check(f'EDMDiscretization n={n_}', OurEDM()(n_), RefEDM()(n_), 0)
        check(f'EDMDiscretization n={n_} flipped', OurEDM()(n_, do_append_zero=False, flip=True), RefEDM()(n_, do_append_zero=False, flip=True), 0)
    check('EDMDiscretization (0.01, 20, 5)', OurEDM(0.01, 20.0, 5.0)(13), RefEDM(0.01, 20.0, 5.0)(13), 0)

    # LinearMultistepSampler (deterministic, order 4 and 2)
    for order in (4, 2):
        y_ref = run(S.LinearMultistepSampler(order=order, discretization_config=dc, num_steps=steps, guider_config=gc, device='cpu'))
        tr = []
        y_or = osamp.linear_multistep_sample(onet, z.clone(), cond, uc, steps, 6.5, order, tr)
        check(f'LinearMultistep order {order} {steps} steps final latent', y_or, y_ref, 2e-4)
        save(f'lms{order}_tiny_8', final=y_ref, mid=tr[steps // 2], order=np.array(order))


SECTIONS = {'t23d': sec_t23d, 'samplers': sec_samplers, 'i23d': sec_i23d, 'ddim': sec_ddim, 'flow_fixed': sec_flow_fixed,
            'more_samplers': sec_more_samplers}

if __name__ == '__main__':
    from make_golden_render import sec_render, sec_decoder, sec_render_presets   # noqa: E402
    from make_golden_unet import sec_unet   # noqa: E402
    SECTIONS.update(render=sec_render, decoder=sec_decoder, render_presets=sec_render_presets, unet=sec_unet)
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        t0 = time.time()
        SECTIONS[s]()
        print(f'-- {s} done in {time.time() - t0:.1f}s')
