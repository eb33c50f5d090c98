"""GPU parity of the device-resident sampler loops against the reference goldens (final latents).
The loops are chaotic amplifiers of rounding error, so the tolerance is on the FINAL latent relative to its norm:
bf16 network inside an fp32 sampler state: rel-L2 <= 1e-2 after 10..250 steps (measured 1.6e-4 .. 3.2e-4: the samplers
contract, the per-step network error ~1e-3 does not compound)."""
import numpy as np
import pytest
import torch

from conftest import golden, load_synth, rel_l2

pytestmark = pytest.mark.gpu


def _tiny():
    from ln3diff_amd.dit.dit_trilatent import DiT_TriLatent
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    m = DiT_TriLatent(input_size=32, patch_size=2, in_channels=4, hidden_size=128, depth=2, num_heads=2, num_classes=0,
                      learn_sigma=False, context_dim=768, roll_out=True, vit_blk=TextCondDiTBlock)
    load_synth(m, 0)
    return m.cuda()


@pytest.mark.parametrize("steps", [10, 250])
def test_edm_euler_cfg_vs_reference_golden(hip_lib, steps):
    from ln3diff_amd.sgm.sampling import EulerEDMSampler, DiscreteDenoiser, VanillaCFG
    from ln3diff_amd.synth import synth_input
    g = golden(f'edm_tiny_{steps}')
    m = _tiny()
    z = synth_input('z', (2, 12, 32, 32), 41).cuda()
    cond = {'crossattn': synth_input('c', (2, 77, 768), 41).cuda()}
    uc = {'crossattn': torch.zeros_like(cond['crossattn'])}
    sampler = EulerEDMSampler(num_steps=steps, guider=VanillaCFG(6.5))
    # host libm/SIMD differences between machines move the fp32 table by an ulp: compare to 1e-6 relative
    assert torch.allclose(sampler.discretization(steps), torch.from_numpy(g['sigmas']), rtol=1e-6, atol=0)
    den = DiscreteDenoiser()
    assert den.quantize(sampler.discretization(steps)[0])[1] == int(g['idx_first'][0])
    tr = []
    y = sampler(den.bind(m), z, cond, uc, trace=tr)
    e0, em, e1 = rel_l2(tr[0].cpu(), g['first']), rel_l2(tr[steps // 2].cpu(), g['mid']), rel_l2(y.cpu(), g['final'])
    print('edm', steps, 'first', e0, 'mid', em, 'final', e1)
    assert e0 < 2e-3 and e1 < 1e-2, (e0, em, e1)


def test_edm_euler_with_churn_vs_reference_golden(hip_lib):
    """r6: EDMSampler's noise injection (s_churn, s_tmin, s_tmax, s_noise; sampling.py:82-130): fast path (fused network loop) and generic
    path (opaque closure) against the reference's own sampler, the draws re-created from its seed; the timesteps the network sees are the
    quantised sigma_hat's."""
    from ln3diff_amd.sgm.sampling import EulerEDMSampler, DiscreteDenoiser, VanillaCFG
    from ln3diff_amd.synth import synth_input
    g = golden('edm_tiny_10_churn')
    m = _tiny()
    z = synth_input('z', (2, 12, 32, 32), 41).cuda()
    cond = {'crossattn': synth_input('c', (2, 77, 768), 41).cuda()}
    uc = {'crossattn': torch.zeros_like(cond['crossattn'])}
    torch.manual_seed(int(g['noise_seed']))
    draws = {int(i): torch.randn(2, 12, 32, 32) for i in g['churned']}
    kw = dict(s_churn=float(g['s_churn']), s_tmin=float(g['s_tmin']), s_tmax=float(g['s_tmax']), s_noise=float(g['s_noise']))
    sampler = EulerEDMSampler(num_steps=10, guider=VanillaCFG(6.5), **kw)
    den = DiscreteDenoiser()
    sig = sampler.discretization(10)
    gam = sampler._gammas(sig)
    assert [i for i in range(10) if gam[i] > 0] == [int(i) for i in g['churned']]
    assert [den.quantize(float(sig[i]) * (1 + gam[i]))[1] for i in range(10)] == [int(t) for t in g['idx_seen']]
    tr = []
    y = sampler(den.bind(m), z.clone(), cond, uc, trace=tr, step_noise=lambda i: draws[i])

    class Opaque:
        def __call__(self, input, sigma, c):
            return den(m, input, sigma, c)
    yg = sampler(Opaque(), z.clone(), cond, uc, step_noise=lambda i: draws[i])
    e, em, eg = rel_l2(y.cpu(), g['final']), rel_l2(tr[5].cpu(), g['mid']), rel_l2(yg.cpu(), g['final'])
    print('edm churn: fast', e, 'mid', em, 'generic', eg)
    assert e < 1e-2 and em < 1e-2 and eg < 1e-2
    assert rel_l2(yg, y) < 1e-4
    y0 = EulerEDMSampler(num_steps=10, guider=VanillaCFG(6.5))(den.bind(m), z.clone(), cond, uc)
    assert rel_l2(y, y0) > 1e-2                                      # the injection is not a no-op


def test_other_sgm_samplers_vs_reference_goldens(hip_lib):
    """r6: HeunEDMSampler (with / without churn), EulerAncestralSampler (two eta / s_noise settings), DPMPP2SAncestralSampler,
    DPMPP2MSampler, LinearMultistepSampler (sampling.py:133-365) on the HIP path against the reference's own classes; the stochastic ones with the reference's RNG
    stream re-drawn from the stored seed; through a bound DiscreteDenoiser, an opaque closure and network=."""
    from ln3diff_amd.sgm import sampling as S
    from ln3diff_amd.synth import synth_input
    m = _tiny()
    z = synth_input('z', (2, 12, 32, 32), 41).cuda()
    cond = {'crossattn': synth_input('c', (2, 77, 768), 41).cuda()}
    uc = {'crossattn': torch.zeros_like(cond['crossattn'])}
    den = S.DiscreteDenoiser()
    cfg = dict(num_steps=8, guider=S.VanillaCFG(6.5))
    res = {}
    g = golden('heun_tiny_8')
    tr = []
    y = S.HeunEDMSampler(**cfg)(den.bind(m), z.clone(), cond, uc, trace=tr)
    res['heun'] = (rel_l2(y.cpu(), g['final']), rel_l2(tr[4].cpu(), g['mid']))
    y2 = S.HeunEDMSampler(**cfg)(den, z.clone(), cond, uc, network=m)
    assert rel_l2(y2, y) < 1e-5
    g = golden('heun_tiny_8_churn')
    torch.manual_seed(int(g['noise_seed']))
    draws = {int(i): torch.randn(2, 12, 32, 32) for i in g['churned']}
    kw = dict(s_churn=float(g['s_churn']), s_tmin=float(g['s_tmin']), s_tmax=float(g['s_tmax']), s_noise=float(g['s_noise']))
    y = S.HeunEDMSampler(**cfg, **kw)(den.bind(m), z.clone(), cond, uc, step_noise=lambda i: draws[i])
    res['heun_churn'] = (rel_l2(y.cpu(), g['final']),)
    for tag, cls in (('euler_ancestral_tiny_8', S.EulerAncestralSampler), ('euler_ancestral_tiny_8_eta', S.EulerAncestralSampler),
                     ('dpmpp2s_tiny_8', S.DPMPP2SAncestralSampler)):
        g = golden(tag)
        torch.manual_seed(int(g['noise_seed']))
        dr = [torch.randn(2, 12, 32, 32) for _ in range(8)]
        tr = []

        class Opaque:
            def __call__(self, input, sigma, c):
                return den(m, input, sigma, c)
        y = cls(eta=float(g['eta']), s_noise=float(g['s_noise']), **cfg)(Opaque(), z.clone(), cond, uc, trace=tr, step_noise=lambda i: dr[i])
        res[tag] = (rel_l2(y.cpu(), g['final']), rel_l2(tr[4].cpu(), g['mid']))
    g = golden('dpmpp2m_tiny_8')
    tr = []
    y = S.DPMPP2MSampler(**cfg)(den.bind(m), z.clone(), cond, uc, trace=tr)
    res['dpmpp2m'] = (rel_l2(y.cpu(), g['final']), rel_l2(tr[4].cpu(), g['mid']))
    for order in (4, 2):
        g = golden('lms%d_tiny_8' % order)
        tr = []
        y = S.LinearMultistepSampler(order=order, **cfg)(den.bind(m), z.clone(), cond, uc, trace=tr)
        res['lms%d' % order] = (rel_l2(y.cpu(), g['final']), rel_l2(tr[4].cpu(), g['mid']))
    # IdentityGuider (the reference's default guider): B rows through the network instead of 2B, the same loops
    for tag, cls in (('euler_identity_tiny_8', S.EulerEDMSampler), ('heun_identity_tiny_8', S.HeunEDMSampler)):
        g = golden(tag)
        seen = []

        class Count:
            def __call__(self, input, sigma, c):
                seen.append(input.shape[0])
                return den(m, input, sigma, c)
        y = cls(num_steps=8, guider=S.IdentityGuider())(Count(), z.clone(), cond, None)
        assert seen and all(b == 2 for b in seen) and len(seen) == int(g['net_calls'])
        res[tag] = (rel_l2(y.cpu(), g['final']),)
    y = S.EulerEDMSampler(num_steps=8, guider=S.IdentityGuider())(den.bind(m), z.clone(), cond, None)      # a bound pair: still the un-doubled loop
    assert rel_l2(y.cpu(), golden('euler_identity_tiny_8')['final']) < 1e-2
    # the other denoiser scalings and the continuous Denoiser (denoiser.py:13-78, denoiser_scaling.py:14-59): the network must see the reference's
    # noise labels (table indices / 0.25 log sigma) and the sampler must leave the fused Eps loop
    for tag, dn in (('euler_vscaling_tiny_8', S.DiscreteDenoiser(scaling=S.VScaling())),
                    ('euler_vscaling_edmcnoise_tiny_8', S.DiscreteDenoiser(scaling=S.VScalingWithEDMcNoise(), quantize_c_noise=False)),
                    ('euler_edmscaling_cont_tiny_8', S.Denoiser(scaling=S.EDMScaling()))):
        g = golden(tag)
        labels = []

        class Net:                                    # an opaque network wrapper: records the noise label of every call
            def __call__(self, x, t, c, **kw):
                labels.append(float(t[0]))
                return m(x, t, c)
        y = S.EulerEDMSampler(**cfg)(dn.bind(Net()), z.clone(), cond, uc)
        res[tag] = (rel_l2(y.cpu(), g['final']),)
        assert np.allclose(np.array(labels, dtype=np.float32), g['labels'], rtol=1e-5, atol=1e-6), (tag, labels[:3], g['labels'][:3])
        y2 = S.EulerEDMSampler(**cfg)(dn, z.clone(), cond, uc, network=m)                  # network=: the package's DiT, still the generic loop
        assert rel_l2(y2.cpu(), g['final']) < 1e-2
    print('other samplers vs reference:', {k: tuple(round(v, 5) for v in e) for k, e in res.items()})
    for k, e in res.items():
        assert max(e) < 1e-2, (k, e)
    # a device draw where no stream is given: finite, and different from the deterministic Euler result
    y = S.EulerAncestralSampler(**cfg)(den.bind(m), z.clone(), cond, uc)
    assert torch.isfinite(y).all()


def test_config1_ditb2_ddpm50_vs_reference_golden(hip_lib):
    """BASELINE config 1: DiT-B/2, SpacedDiffusion('50').p_sample_loop, B=1 - the reference's CPU-runnable case."""
    from ln3diff_amd.dit.dit_trilatent import DiT_models
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    from ln3diff_amd.guided_diffusion import gaussian_diffusion as gd
    from ln3diff_amd.guided_diffusion.respace import SpacedDiffusion, space_timesteps
    from ln3diff_amd.synth import synth_input
    g = golden('config1_ditb2_ddpm50')
    m = DiT_models['DiT-B/2'](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=768,
                              roll_out=True, vit_blk=TextCondDiTBlock)
    load_synth(m, 0)
    m = m.cuda()
    diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, '50'), betas=gd.get_named_beta_schedule('linear', 1000),
                           model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_LARGE)
    ctx = synth_input('ctx', (1, 77, 768), 1).cuda()
    z = synth_input('z', (1, 12, 32, 32), 1).cuda()
    torch.manual_seed(int(g['noise_seed']))
    noises = [torch.randn(1, 12, 32, 32) for _ in range(50)]       # the reference's randn_like stream
    tr = []
    y = diff.p_sample_loop(m, (1, 12, 32, 32), cond=ctx, noise=z, clip_denoised=False, mixing_normal=False,
                           step_noise=lambda k: noises[k], trace=tr)
    e0, e24, e = rel_l2(tr[0].cpu(), g['step0']), rel_l2(tr[24].cpu(), g['step24']), rel_l2(y.cpu(), g['final'])
    print('config1 step0', e0, 'step24', e24, 'final', e)
    assert e0 < 2e-3 and e < 1e-2, (e0, e24, e)


def test_ddpm250_ditl2_vs_reference_golden(hip_lib):
    """north_star's "250-step DDPM" at the benchmark's model size: DiT-L/2 through SpacedDiffusion('250').p_sample_loop, B = 1
    (reference loop on the CPU, tests/golden/make_golden_geom.py ddpm250_l2): steps 0 / 124 / final."""
    from ln3diff_amd.dit.dit_trilatent import DiT_models
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    from ln3diff_amd.guided_diffusion import gaussian_diffusion as gd
    from ln3diff_amd.guided_diffusion.respace import SpacedDiffusion, space_timesteps
    from ln3diff_amd.synth import synth_input
    g = golden('ddpm250_ditl2')
    m = DiT_models['DiT-L/2'](input_size=32, num_classes=0, learn_sigma=False, in_channels=4, context_dim=768,
                              roll_out=True, vit_blk=TextCondDiTBlock)
    load_synth(m, 0)
    m = m.cuda()
    diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, '250'), betas=gd.get_named_beta_schedule('linear', 1000),
                           model_mean_type=gd.ModelMeanType.EPSILON, model_var_type=gd.ModelVarType.FIXED_LARGE)
    ctx = synth_input('ctx', (1, 77, 768), 1).cuda()
    z = synth_input('z', (1, 12, 32, 32), 1).cuda()
    torch.manual_seed(int(g['noise_seed']))
    noises = [torch.randn(1, 12, 32, 32) for _ in range(250)]      # the reference's randn_like stream
    tr = []
    y = diff.p_sample_loop(m, (1, 12, 32, 32), cond=ctx, noise=z, clip_denoised=False, mixing_normal=False,
                           step_noise=lambda k: noises[k], trace=tr)
    e0, e124, e = rel_l2(tr[0].cpu(), g['step0']), rel_l2(tr[124].cpu(), g['step124']), rel_l2(y.cpu(), g['final'])
    print('DDPM-250 DiT-L/2 step0', e0, 'step124', e124, 'final', e)
    assert e0 < 2e-3 and e < 1e-2, (e0, e124, e)


@pytest.mark.parametrize("spec", ['ddim50', 'ddim25'])
def test_ddim_cfg_vs_reference_golden(hip_lib, spec):
    from ln3diff_amd.guided_diffusion import gaussian_diffusion as gd
    from ln3diff_amd.guided_diffusion.respace import SpacedDiffusion, space_timesteps
    from ln3diff_amd.synth import synth_input
    g = golden(f'ddim_tiny_{spec}')
    m = _tiny()
    diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, spec), betas=gd.get_named_beta_schedule('linear', 1000))
    z = synth_input('z', (2, 12, 32, 32), 41).cuda()
    c = synth_input('c', (2, 77, 768), 41).cuda()
    torch.manual_seed(int(g['noise_seed']))
    noises = [torch.randn(2, 12, 32, 32) for _ in range(diff.num_timesteps)]
    y = diff.ddim_sample_loop(m, (2, 12, 32, 32), cond={'c_crossattn': c}, noise=z, clip_denoised=False, eta=float(g['eta']),
                              unconditional_guidance_scale=float(g['scale']), step_noise=lambda k: noises[k])
    e = rel_l2(y.cpu(), g['final'])
    print('ddim', spec, e)
    assert e < 1e-2, e


def test_edm_graph_replay_is_bitwise_identical(hip_lib):
    """EulerEDMSampler(use_graph=True): the network evaluation captured in a HIP graph and replayed per step gives the same bits as
    the launch-by-launch loop."""
    from ln3diff_amd.sgm.sampling import EulerEDMSampler, DiscreteDenoiser, VanillaCFG
    from ln3diff_amd.synth import synth_input
    m = _tiny()
    z = synth_input('z', (2, 12, 32, 32), 41).cuda()
    cond = {'crossattn': synth_input('c', (2, 77, 768), 41).cuda()}
    uc = {'crossattn': torch.zeros_like(cond['crossattn'])}
    ya = EulerEDMSampler(num_steps=12, guider=VanillaCFG(6.5), use_graph=False)(DiscreteDenoiser().bind(m), z.clone(), cond, uc)
    yb = EulerEDMSampler(num_steps=12, guider=VanillaCFG(6.5), use_graph=True)(DiscreteDenoiser().bind(m), z.clone(), cond, uc)
    assert torch.isfinite(yb).all() and torch.equal(ya, yb)
