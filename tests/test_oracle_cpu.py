"""CPU suite: the oracle against the committed golden vectors (produced by the REFERENCE in the build
container, tests/golden/make_golden.py), host logic, and that the C-ABI library loads and exports every
symbol of include/ln3d.h.  No compute calls into the HIP library here (no GPU)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, load_synth, manifest, rel_l2
from ln3diff_amd.synth import synth_input, synth_state_dict, orbit_cameras
from oracle import dit as odit, samplers as osamp, render as orender, decoder as odec


def _sd_from_manifest(g, key='manifest'):
    shapes = manifest(g, key)
    computed = {}
    for k, s in shapes.items():
        if k == 'pos_embed':
            computed[k] = odit.trilatent_pos_embed(s[-1])
        elif k.endswith('vit_decoder.pos_embed'):
            computed[k] = odec.decoder_pos_embed(s[-1])
    return synth_state_dict(shapes, 0, computed)


def test_abi_library_loads_and_exports_header_symbols(hip_lib):
    from ln3diff_amd import _lib
    assert _lib.check_symbols()
    hdr = open(os.path.join(ROOT, 'include', 'ln3d.h')).read()
    declared = set(re.findall(r'\b(ln3d_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'ln3d_gemm_args', 'ln3d_attn_args', 'ln3d_norm_args', 'ln3d_render_args'}
    for s in declared:
        assert hasattr(hip_lib, s), s
    assert declared <= set(_lib.SYMBOLS) | {'ln3d_strerror'}
    # the ctypes stub of INTEGRATION.md pins the same ABI number the library reports (its argument structs are checked on the GPU)
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    assert [int(v) for v in re.findall(r'ln3d_abi_version\(\) == (\d+)', doc)] == [hip_lib.ln3d_abi_version()]


def test_product_fails_loudly_without_gpu():
    from ln3diff_amd.dit.dit_trilatent import DiT_TriLatent
    from ln3diff_amd.dit.dit_models_xformers import TextCondDiTBlock
    m = DiT_TriLatent(hidden_size=128, depth=1, num_heads=2, num_classes=0, learn_sigma=False, context_dim=768,
                      roll_out=True, vit_blk=TextCondDiTBlock)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 12, 32, 32), torch.zeros(1), torch.zeros(1, 77, 768))


def test_oracle_t23d_tiny_matches_reference_golden():
    g = golden('t23d_tiny')
    sd = _sd_from_manifest(g)
    x = synth_input('x', (2, 12, 32, 32), 0)
    ctx = synth_input('ctx', (2, 77, 768), 0)
    y = odit.t23d_forward(sd, x, torch.from_numpy(g['t']), ctx, 2)
    assert rel_l2(y, g['y']) < 1e-5


def test_oracle_i23d_tiny_matches_reference_golden():
    g = golden('i23d_tiny')
    sd = _sd_from_manifest(g)
    x = synth_input('x', (4, 12, 32, 32), 0)
    ctx = {'crossattn': synth_input('ca', (4, 256, 2048), 0), 'vector': synth_input('v', (4, 768), 0)}
    y = odit.i23d_forward_with_cfg(sd, x, torch.from_numpy(g['t']), ctx, 4.0, 2)
    assert rel_l2(y, g['y']) < 1e-5


@pytest.mark.parametrize("tag,hidden,depth,heads,patch", [("tiny", 128, 2, 2, 2), ("h72", 1152, 1, 16, 2), ("p1", 128, 1, 2, 1)])
def test_oracle_i23d_plain_matches_reference_golden(tag, hidden, depth, heads, patch):
    """plain DiT_I23D (ImageCondDiTBlock blocks, dit/dit_i23d.py:24-170): the host mirror has the reference's state-dict manifest
    and the oracle reproduces the reference output."""
    from ln3diff_amd.dit.dit_i23d import DiT_I23D, DiT_models
    g = golden('i23d_plain_' + tag)
    m = DiT_I23D(input_size=32, patch_size=patch, in_channels=4, hidden_size=hidden, depth=depth, num_heads=heads, num_classes=0,
                 learn_sigma=False, context_dim=1024, roll_out=True)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == manifest(g)
    sd, _ = load_synth(m, 0)
    ctx = {'crossattn': synth_input('ca', (2, 256, 2048), 9), 'vector': synth_input('v', (2, 1024), 9)}
    y = odit.i23d_plain_forward(sd, synth_input('x', (2, 12, 32, 32), 9), torch.from_numpy(g['t']), ctx, heads, patch)
    assert rel_l2(y, g['y']) < 1e-5
    assert {'DiT-XL/2', 'DiT-L/2', 'DiT-B/2', 'DiT-B/1'} <= set(DiT_models)          # dit_i23d.py:686-690


def test_oracle_edm_euler_matches_reference_golden():
    g = golden('edm_tiny_10')
    sd = _sd_from_manifest(golden('t23d_tiny'))
    z = synth_input('z', (2, 12, 32, 32), 41)
    cond = {'crossattn': synth_input('c', (2, 77, 768), 41), 'vector': synth_input('v', (2, 768), 41)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    assert torch.equal(osamp.legacy_ddpm_sigmas(10), torch.from_numpy(g['sigmas']))
    tr = []
    y = osamp.edm_euler_sample(lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2), z, cond, uc, 10, 6.5, tr)
    assert rel_l2(tr[0], g['first']) < 1e-5 and rel_l2(y, g['final']) < 1e-4
    s250 = osamp.legacy_ddpm_sigmas(250)
    assert abs(float(s250[0]) - 14.6146) < 1e-3 and abs(float(s250[249]) - 0.0586) < 1e-3 and float(s250[250]) == 0


def test_oracle_edm_euler_with_churn_matches_reference_golden():
    """r6: s_churn / s_tmin / s_tmax / s_noise (sampling.py:82-130) through the reference's own sampler, its RNG stream re-drawn from the seed."""
    g = golden('edm_tiny_10_churn')
    sd = _sd_from_manifest(golden('t23d_tiny'))
    z = synth_input('z', (2, 12, 32, 32), 41)
    cond = {'crossattn': synth_input('c', (2, 77, 768), 41), 'vector': synth_input('v', (2, 768), 41)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    torch.manual_seed(int(g['noise_seed']))
    draws = {int(i): torch.randn(2, 12, 32, 32) for i in g['churned']}
    y = osamp.edm_euler_sample(lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2), z, cond, uc, 10, 6.5, None, step_noise=lambda i: draws[i],
                               s_churn=float(g['s_churn']), s_tmin=float(g['s_tmin']), s_tmax=float(g['s_tmax']), s_noise=float(g['s_noise']))
    assert 0 < len(draws) < 10 and rel_l2(y, g['final']) < 1e-4


@pytest.mark.parametrize("method,steps", [('heun', 10), ('midpoint', 10), ('rk4', 6)])
def test_oracle_flow_fixed_grid_matches_reference_golden(method, steps):
    g = golden(f'flow_tiny_{method}{steps}')
    sd = _sd_from_manifest(golden('i23d_tiny'))
    z = synth_input('z', (2, 12, 32, 32), 42)
    cond = {'crossattn': synth_input('ca', (2, 256, 2048), 42), 'vector': synth_input('v', (2, 768), 42)}
    ctx = {k: torch.cat([v, torch.zeros_like(v)], 0) for k, v in cond.items()}
    y = osamp.flow_ode_sample(lambda x, t, **kw: odit.i23d_forward_with_cfg(sd, x, t, kw['context'], 4.0, 2),
                              torch.cat([z, z]), steps, method, context=ctx).chunk(2)[0]
    assert rel_l2(y, g['final']) < 1e-4


@pytest.mark.parametrize("method,steps,form,last", [('Euler', 25, 'sigma', 'Mean'), ('Heun', 8, 'linear', 'Euler'),
                                                    ('Euler', 12, 'decreasing', 'Tweedie')])
def test_oracle_flow_sde_matches_reference_golden(method, steps, form, last):
    g = golden(f'sde_tiny_{method.lower()}{steps}_{form}_{last.lower()}')
    sd = _sd_from_manifest(golden('i23d_tiny'))
    z = synth_input('z', (2, 12, 32, 32), 42)
    cond = {'crossattn': synth_input('ca', (2, 256, 2048), 42), 'vector': synth_input('v', (2, 768), 42)}
    ctx = {k: torch.cat([v, torch.zeros_like(v)], 0) for k, v in cond.items()}
    torch.manual_seed(1234)
    y = osamp.flow_sde_sample(lambda x, t, **kw: odit.i23d_forward_with_cfg(sd, x, t, kw['context'], 4.0, 2),
                              torch.cat([z, z]), steps, method, form, 0.7, last, 0.04, context=ctx)[-1].chunk(2)[0]
    assert rel_l2(y, g['final']) < 1e-4
    with pytest.raises(TypeError):          # the reference's 'constant' form is broken (sqrt of a float): same error here
        osamp.flow_sde_sample(lambda x, t, **kw: x, torch.zeros(1, 2), 3, 'Euler', 'constant')


def test_schedule_tables():
    assert osamp.space_timesteps(1000, 'ddim250') == set(range(0, 1000, 4))
    s50 = sorted(osamp.space_timesteps(1000, '50'))
    assert s50[0] == 0 and s50[-1] == 999 and len(s50) == 50
    tb = osamp.SpacedTables('50')
    assert tb.timestep_map == s50 and abs(np.prod(1 - tb.betas) - np.cumprod(1 - osamp.linear_betas())[-1]) < 1e-12


def _dec_sd(bias):
    shapes = {'net.0.weight': (64, 32), 'net.0.bias': (64,), 'net.2.weight': (4, 64), 'net.2.bias': (4,)}
    sd = synth_state_dict(shapes, 0)
    sd['net.2.bias'] = sd['net.2.bias'].clone()
    sd['net.2.bias'][0] += bias
    return sd


@pytest.mark.parametrize('tag,res,V', [('dense_r16', 16, 2), ('sparse_r16', 16, 1)])
def test_oracle_render_matches_reference_golden(tag, res, V):
    from ln3diff_amd.nsr.triplane import draw_render_noise
    g = golden('render_' + tag)
    planes = synth_input('planes', (V, 96, 128, 128), 3, float(g['plane_scale']))
    gen = torch.Generator().manual_seed(int(g['jitter_seed']))
    j, u = draw_render_noise(V, res * res, 64, generator=gen)
    r = orender.triplane_render(planes, _dec_sd(float(g['sigma_bias'])), torch.from_numpy(g['cams']), res,
                                j.unsqueeze(-1), u)
    for k in ('image_raw', 'image_depth', 'weights_samples', 'image_mask'):
        assert rel_l2(r[k], g[k]) < 1e-4, k
    if tag.startswith('sparse'):     # the batch-global depth clamp quirk: every pixel == global min depth
        assert float(r['image_depth'].max() - r['image_depth'].min()) == 0.0
    assert np.allclose(orbit_cameras(8)[[1, 6][:V]].numpy(), g['cams'])


@pytest.mark.parametrize('tag,optname', [('shapenet64', 'SHAPENET_OPTS'), ('objv96', 'OBJAVERSE_96_OPTS'), ('afhq48', 'AFHQ_48_OPTS')])
def test_oracle_render_presets_match_reference_golden(tag, optname):
    """The other sampling presets (numeric ray limits, 48 / 96 samples, no bbox filter, black background) and the seam outputs of
    ImportanceRenderer.forward (visibility, return_meta's weights / all_coords / feature_volume) against the reference's outputs."""
    from ln3diff_amd.nsr.triplane import draw_render_noise
    g = golden('render_preset_' + tag)
    opts = getattr(orender, optname)
    cams = torch.from_numpy(g['cams'])
    res, V = int(g['res']), cams.shape[0]
    M, S, NI = res * res, opts['depth_resolution'], opts['depth_resolution_importance']
    planes = synth_input('planes', (V, 96, 128, 128), 3, float(g['plane_scale']))
    gen = torch.Generator().manual_seed(int(g['jitter_seed']))
    if opts['ray_start'] == 'auto':
        j, u = draw_render_noise(V, M, S, generator=gen, n_importance=NI)
        j = j.unsqueeze(-1)
    else:
        j, u = torch.rand(V, M, S, 1, generator=gen), None
        u = torch.rand(V * M, NI, generator=gen)
    r = orender.triplane_render(planes, _dec_sd(float(g['sigma_bias'])), cams, res, j, u, opts)
    for k in ('image_raw', 'image_depth', 'weights_samples', 'image_mask'):
        assert rel_l2(r[k], g[k]) < 1e-4, k
    d = r['detail']
    assert rel_l2(d['visibility'], g['visibility']) < 1e-4
    for k in ('weights', 'all_coords', 'feature_volume'):
        assert rel_l2(d[k], g[k].astype(np.float32)) < 2e-3, k           # stored as fp16


def test_oracle_grid_matches_reference_golden():
    g = golden('grid16')
    planes = synth_input('planes', (1, 96, 128, 128), 3, 4.0)
    r = orender.decode_grid(planes, _dec_sd(4.0), 16)
    assert rel_l2(r['sigma'][0, ..., 0], g['sigma']) < 1e-5 and rel_l2(r['rgb'][0], g['rgb']) < 1e-5


def test_oracle_decode_tiny_matches_reference_golden():
    g = golden('decode_tiny')
    sd = _sd_from_manifest(g)
    latent = synth_input('latent', (2, 12, 32, 32), 5)
    planes = odec.vae_decode(sd, latent, 2)
    assert rel_l2(planes[:, :, ::8, ::8], g['planes_sub']) < 1e-4
    assert abs(float(planes.std()) - float(g['planes_std'])) < 1e-4


def test_oracle_ddim_matches_reference_golden():
    g = golden('ddim_tiny_ddim25')
    sd = _sd_from_manifest(golden('t23d_tiny'))
    z = synth_input('z', (2, 12, 32, 32), 41)
    c = synth_input('c', (2, 77, 768), 41)
    torch.manual_seed(int(g['noise_seed']))
    noises = [torch.randn(2, 12, 32, 32) for _ in range(25)]
    y = osamp.ddim_sample_loop(lambda x, t, cc: odit.t23d_forward(sd, x, t, cc, 2), z, c, osamp.SpacedTables('ddim25'),
                               float(g['eta']), float(g['scale']), None, noises)
    assert rel_l2(y, g['final']) < 1e-4


@pytest.mark.parametrize("name", ["tiny", "tiny_eos"])
def test_oracle_clip_text_matches_transformers_golden(name):
    import json
    from oracle import clip_text as oclip
    g = golden(f'clip_text_{name}')
    shapes = {k: tuple(v) for k, v in json.loads(str(g['manifest'])).items()}
    sd = synth_state_dict(shapes, 0)
    last, pooled = oclip.clip_text_forward(sd, torch.from_numpy(g['ids']).long(), int(g['heads']), int(g['eos_token_id']))
    assert rel_l2(last, g['last']) < 1e-5 and rel_l2(pooled, g['pooled']) < 1e-5


def test_oracle_i23d_multiview_matches_reference_golden():
    g = golden('i23d_mv_tiny')
    sd = _sd_from_manifest(g)
    ctx = {'crossattn': synth_input('ca', (2, 256, 1024), 5), 'vector': synth_input('v', (2, 768), 5),
           'concat': synth_input('mv', (2, 4, 256, 768), 5)}
    y = odit.i23d_mv_forward(sd, synth_input('x', (2, 12, 32, 32), 5), torch.from_numpy(g['t']), ctx, 2)
    assert rel_l2(y, g['y']) < 1e-4


def test_oracle_image_towers_match_transformers_goldens():
    import json
    from oracle import vit_image as ovit
    from ln3diff_amd.synth import synth_vit_state_dict
    g = golden('vit_clip_tiny')
    sh = {k: tuple(v) for k, v in json.loads(str(g['manifest'])).items()}
    pooled, tokens, _ = ovit.openclip_visual_forward(synth_state_dict(sh, 0), synth_input('img', (2, 3, 56, 56), 3), int(g['heads']))
    assert rel_l2(pooled, g['pooled']) < 1e-5 and rel_l2(tokens[:, ::int(g['tok_stride'])], g['tokens']) < 1e-5
    g = golden('vit_dino_tiny')
    sh = {k: tuple(v) for k, v in json.loads(str(g['manifest'])).items()}
    cls, tokens = ovit.dinov2_forward(synth_vit_state_dict(sh, 0), synth_input('img', (2, 3, 56, 56), 4), int(g['heads']))
    assert rel_l2(cls, g['cls']) < 1e-5 and rel_l2(tokens[:, ::int(g['tok_stride'])], g['tokens']) < 1e-5


def test_oracle_i23d_multiview_noclip_matches_reference_golden():
    g = golden('i23d_mv_noclip_tiny')
    sd = _sd_from_manifest(g)
    y = odit.i23d_mv_noclip_forward(sd, synth_input('x', (2, 12, 32, 32), 5), torch.from_numpy(g['t']),
                                    {'concat': synth_input('mv', (2, 4, 256, 768), 5)}, 2)
    assert rel_l2(y, g['y']) < 1e-4


def test_oracle_multiview_conditioner_matches_goldens():
    """r4 (f)3: Pluecker ray maps against the REFERENCE's own get_plucker_ray (fixture of tests/golden/make_golden_mv.py, sub-sampled),
    the 9-channel DINOv2-reg conditioner against the transformers-pinned fixture at the tiny size."""
    import json
    from oracle import vit_image as ovit
    from ln3diff_amd.synth import synth_vit_state_dict
    g = golden('mv_plucker_rays')
    st = int(g['stride'])
    rays = ovit.plucker_rays(torch.from_numpy(g['c']), 224)
    assert rays.shape == (6, 6, 224, 224) and rel_l2(rays[:, :, ::st, ::st], g['rays']) < 1e-6
    g = golden('mv_plucker_tiny')
    sh = {k: tuple(v) for k, v in json.loads(str(g['manifest'])).items()}
    T, S = int(g['n_cond_frames']), int(g['size'])
    img_c = {'img': synth_input('mvimg', (2, T + 1, 3, S, S), 7).clamp(-1, 1), 'c': torch.from_numpy(g['c'])}
    tok = ovit.dinov2_mv_plucker_forward(synth_vit_state_dict(sh, 0), img_c, int(g['heads']), n_cond_frames=T, size=S)
    assert tok.shape[:2] == (2, T) and rel_l2(tok[:, :, ::int(g['tok_stride'])], g['tokens']) < 1e-5


def test_oracle_other_sgm_samplers_match_reference_goldens():
    """r6: Heun (with / without churn), Euler-ancestral (two eta / s_noise settings), DPM++ 2S ancestral, DPM++ 2M, linear multistep (sampling.py:133-365)
    through the reference's own classes (tests/golden/make_golden.py::sec_more_samplers), the stochastic ones with the reference's RNG
    stream re-drawn from the stored seed."""
    sd = _sd_from_manifest(golden('t23d_tiny'))
    z = synth_input('z', (2, 12, 32, 32), 41)
    cond = {'crossattn': synth_input('c', (2, 77, 768), 41), 'vector': synth_input('v', (2, 768), 41)}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    net = lambda x, t, c: odit.t23d_forward(sd, x, t, c, 2)
    g = golden('heun_tiny_8')
    tr = []
    y = osamp.edm_heun_sample(net, z.clone(), cond, uc, 8, 6.5, tr)
    assert rel_l2(y, g['final']) < 1e-4 and rel_l2(tr[4], g['mid']) < 1e-4
    g = golden('heun_tiny_8_churn')
    torch.manual_seed(int(g['noise_seed']))
    draws = {int(i): torch.randn(2, 12, 32, 32) for i in g['churned']}
    y = osamp.edm_heun_sample(net, z.clone(), cond, uc, 8, 6.5, None, step_noise=lambda i: draws[i],
                              s_churn=float(g['s_churn']), s_tmin=float(g['s_tmin']), s_tmax=float(g['s_tmax']), s_noise=float(g['s_noise']))
    assert rel_l2(y, g['final']) < 1e-4
    for tag, fn in (('euler_ancestral_tiny_8', osamp.euler_ancestral_sample), ('euler_ancestral_tiny_8_eta', osamp.euler_ancestral_sample),
                    ('dpmpp2s_tiny_8', osamp.dpmpp2s_ancestral_sample)):
        g = golden(tag)
        torch.manual_seed(int(g['noise_seed']))
        draws = [torch.randn(2, 12, 32, 32) for _ in range(8)]
        y = fn(net, z.clone(), cond, uc, 8, 6.5, eta=float(g['eta']), s_noise=float(g['s_noise']), step_noise=lambda i: draws[i])
        assert rel_l2(y, g['final']) < 1e-4, tag
    g = golden('dpmpp2m_tiny_8')
    assert rel_l2(osamp.dpmpp2m_sample(net, z.clone(), cond, uc, 8, 6.5), g['final']) < 1e-4
    assert rel_l2(osamp.edm_euler_sample(net, z.clone(), cond, cond, 8, 1.0), golden('euler_identity_tiny_8')['final']) < 1e-4
    assert rel_l2(osamp.edm_heun_sample(net, z.clone(), cond, cond, 8, 1.0), golden('heun_identity_tiny_8')['final']) < 1e-4
    for tag, kw in (('euler_vscaling_tiny_8', dict(scaling='v')), ('euler_vscaling_edmcnoise_tiny_8', dict(scaling='v_edm', quantize_c_noise=False)),
                    ('euler_edmscaling_cont_tiny_8', dict(scaling='edm', discrete=False))):
        assert rel_l2(osamp.edm_euler_sample(net, z.clone(), cond, uc, 8, 6.5, **kw), golden(tag)['final']) < 1e-4, tag
    for order in (4, 2):
        g = golden('lms%d_tiny_8' % order)
        assert rel_l2(osamp.linear_multistep_sample(net, z.clone(), cond, uc, 8, 6.5, order), g['final']) < 1e-4, order


def test_linear_multistep_weights_equal_the_reference_quadrature():
    """The closed-form Lagrange integrals (oracle and product) against scipy's adaptive quadrature of the same integrand, which is how
    sampling_utils.py:7-19 computes them; every (order, step, j) of an 8- and a 50-step LegacyDDPM schedule."""
    from scipy import integrate
    from ln3diff_amd.sgm.sampling import linear_multistep_coeff as prod_coeff
    for n in (8, 50):
        t = osamp.legacy_ddpm_sigmas(n).numpy()
        for i in range(n):
            for order in range(1, min(i + 1, 4) + 1):
                for j in range(order):
                    def fn(tau):
                        prod = 1.0
                        for k in range(order):
                            if j != k:
                                prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
                        return prod
                    ref = integrate.quad(fn, t[i], t[i + 1], epsrel=1e-4)[0]
                    a, b = osamp.linear_multistep_coeff(order, t, i, j), prod_coeff(order, t, i, j)
                    assert abs(a - ref) <= 1e-6 * max(1.0, abs(ref)) and a == b, (n, i, order, j, a, b, ref)
