"""GPU parity of the fused HIP ray-marcher against the reference's renderer outputs (golden, produced by the
reference in the build container) and the CPU oracle.  fp32 end to end; tolerance rel-L2 <= 2e-3 on images
(fast exp/log in the MLP, different summation order in scans) and exact-ish depth."""
import numpy as np
import pytest
import torch

from conftest import golden, rel_l2

pytestmark = pytest.mark.gpu


def _decoder_sd(sigma_bias):
    from ln3diff_amd.synth import synth_state_dict
    shapes = {'net.0.weight': (64, 32), 'net.0.bias': (64,), 'net.2.weight': (4, 64), 'net.2.bias': (4,)}
    sd = synth_state_dict(shapes, 0)
    sd['net.2.bias'] = sd['net.2.bias'].clone()
    sd['net.2.bias'][0] += sigma_bias
    return sd


@pytest.mark.parametrize("tag,res,V", [('dense_r16', 16, 2), ('dense_r32', 32, 2), ('sparse_r16', 16, 1)])
def test_render_vs_reference_golden(hip_lib, tag, res, V):
    from ln3diff_amd.nsr.triplane import Triplane, draw_render_noise
    from ln3diff_amd.synth import synth_input
    g = golden('render_' + tag)
    tp = Triplane(img_resolution=res)
    tp.decoder.load_state_dict(_decoder_sd(float(g['sigma_bias'])))
    tp = tp.cuda()
    planes = synth_input('planes', (V, 96, 128, 128), 3, float(g['plane_scale']))
    cams = torch.from_numpy(g['cams'])
    gen = torch.Generator().manual_seed(int(g['jitter_seed']))
    jitter, u_fine = draw_render_noise(V, res * res, 64, generator=gen)
    out = tp(planes.cuda(), cams.cuda(), jitter=jitter, u_fine=u_fine, return_debug=True)
    cd = out['shape_synthesized']['coarse_densities'].cpu().reshape(-1)
    cd_ref = torch.from_numpy(g['coarse_densities'].astype(np.float32)).reshape(-1)
    inb = cd_ref > -1e30
    assert torch.equal(cd > -1e30, inb)
    assert rel_l2(cd[inb], cd_ref[inb]) < 2e-3                      # fp16-stored golden
    fd = out['shape_synthesized']['fine_depths'].cpu().reshape(-1)
    assert rel_l2(fd, torch.from_numpy(g['fine_depths'].astype(np.float32)).reshape(-1)) < 2e-3
    for key in ('image_raw', 'image_depth', 'weights_samples', 'image_mask'):
        e = rel_l2(out[key].cpu(), g[key])
        print(tag, key, e)
        assert e < 2e-3, (key, e)


def test_grid_query_vs_reference_golden(hip_lib):
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.synth import synth_input
    g = golden('grid16')
    tp = Triplane(img_resolution=16)
    tp.decoder.load_state_dict(_decoder_sd(4.0))
    tp = tp.cuda()
    planes = synth_input('planes', (1, 96, 128, 128), 3, 4.0).cuda()
    pcl = tp.to_channel_last(planes)
    G = 16
    ax = torch.linspace(-0.45, 0.45, G)
    pts = torch.stack(torch.meshgrid(ax, ax, ax, indexing='ij'), dim=-1).reshape(-1, 3).cuda()
    out = tp.query_points(pcl[0], pts)
    assert rel_l2(out['sigma'].cpu().reshape(G, G, G), g['sigma']) < 1e-4
    assert rel_l2(out['rgb'].cpu().reshape(G, G, G, 3), g['rgb']) < 1e-4


def test_render_256_properties(hip_lib):
    """Full-size 256^2 views: size-independent properties (weights in [0,1], white background where the
    accumulated weight is 0, depth inside the global [min,max] clamp, determinism)."""
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.synth import synth_input, orbit_cameras
    tp = Triplane(img_resolution=256)
    tp.decoder.load_state_dict(_decoder_sd(4.0))
    tp = tp.cuda()
    planes = synth_input('planes', (1, 96, 128, 128), 3, 4.0).cuda()
    pcl = tp.to_channel_last(planes)
    cams = orbit_cameras(4).cuda()
    g = torch.Generator(device='cuda').manual_seed(0)
    j = torch.rand(4, 256 * 256, 64, device='cuda', generator=g)
    u = torch.rand(4 * 256 * 256, 64, device='cuda', generator=g)
    idx = torch.zeros(4, dtype=torch.int32, device='cuda')
    a = tp(c=cams, planes_channel_last=pcl, plane_index=idx, jitter=j, u_fine=u)
    b = tp(c=cams, planes_channel_last=pcl, plane_index=idx, jitter=j, u_fine=u)
    assert torch.equal(a['image_raw'], b['image_raw'])
    w = a['weights_samples']
    assert float(w.min()) >= -1e-5 and float(w.max()) <= 1 + 1e-4
    assert torch.isfinite(a['image_raw']).all() and float(a['image_raw'].abs().max()) <= 1.0 + 2e-3
    empty = w < 1e-6
    if empty.any():
        assert float((a['image_raw'] - 1.0).abs()[empty.expand(-1, 3, -1, -1)].max()) < 1e-4
    d = a['image_depth']
    assert float(d.min()) > 0.5 and float(d.max()) < 3.0


def test_render_is_bitwise_repeatable_over_scenes(hip_lib):
    """r4: the rays on which an SLP build of the marcher differs from launch to launch are few and scene-dependent
    (profiles/r4_render_spill.md), so one scene says little: random planes, several orbits and resolutions, each rendered 4 times by
    the shipped library - identical bits (tools/render_repeat_sweep.py is the same sweep as a tool)."""
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.synth import synth_input, orbit_cameras
    tp = Triplane(img_resolution=256)
    tp.decoder.load_state_dict(_decoder_sd(4.0))
    tp = tp.cuda()
    for seed, (V, res, el, rad) in enumerate(((8, 256, 40.0, 1.5), (12, 128, -20.0, 2.2), (2, 512, 5.0, 1.7719), (8, 256, 75.0, 1.3))):
        pcl = tp.to_channel_last(synth_input('planes', (1, 96, 128, 128), 10 + seed, 4.0).cuda())
        cams = orbit_cameras(V, radius=rad, elevation_deg=el).cuda()
        g = torch.Generator(device='cuda').manual_seed(seed)
        j = torch.rand(V, res * res, 64, device='cuda', generator=g)
        u = torch.rand(V * res * res, 64, device='cuda', generator=g)
        idx = torch.zeros(V, dtype=torch.int32, device='cuda')
        f = lambda: tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
        ref = f()
        for _ in range(3):
            o = f()
            for k in ('image_raw', 'image_depth', 'weights_samples'):
                assert torch.equal(ref[k], o[k]), (V, res, el, rad, k)


PRESET_OPTS = {'shapenet64': 'SHAPENET_OPTS', 'objv128': 'OBJAVERSE_128_OPTS', 'objv96': 'OBJAVERSE_96_OPTS', 'eg3d80': 'EG3D_80_OPTS',
               'afhq48': 'AFHQ_48_OPTS', 'objv64_meta': 'OBJAVERSE_OPTS'}


def _preset_scene(tag):
    from oracle import render as orender
    from ln3diff_amd.nsr.triplane import Triplane, draw_render_noise
    from ln3diff_amd.synth import synth_input
    g = golden('render_preset_' + tag)
    opts = dict(getattr(orender, PRESET_OPTS[tag]))
    res = int(g['res'])
    tp = Triplane(img_resolution=res, rendering_kwargs=dict(opts, return_sampling_details_flag=True))
    tp.decoder.load_state_dict(_decoder_sd(float(g['sigma_bias'])))
    tp = tp.cuda()
    cams = torch.from_numpy(g['cams'])
    V, M, S, NI = cams.shape[0], res * res, opts['depth_resolution'], opts['depth_resolution_importance']
    planes = synth_input('planes', (V, 96, 128, 128), 3, float(g['plane_scale']))
    gen = torch.Generator().manual_seed(int(g['jitter_seed']))
    if opts['ray_start'] == 'auto':                                  # RNG order of the two branches of sample_stratified (renderer.py:455-474)
        jitter, u_fine = draw_render_noise(V, M, S, generator=gen, n_importance=NI)
    else:
        jitter = torch.rand(V, M, S, 1, generator=gen).reshape(V, M, S)
        u_fine = torch.rand(V * M, NI, generator=gen)
    return g, tp, opts, planes, cams, jitter, u_fine, (V, M, S, NI, res)


@pytest.mark.parametrize("tag", ['shapenet64', 'objv128', 'objv96', 'eg3d80', 'afhq48'])
def test_render_presets_vs_reference_golden(hip_lib, tag):
    """The other sampling presets of nsr/script_util.py:433-1000 (numeric ray_start / ray_end, 48 - 128 samples per pass, no bbox
    filter, black background) through Triplane.forward, against the reference's own outputs (tests/golden/make_golden_render.py
    sec_render_presets; oracle == reference <= 1e-4 there)."""
    g, tp, opts, planes, cams, jitter, u_fine, (V, M, S, NI, res) = _preset_scene(tag)
    out = tp(planes.cuda(), cams.cuda(), jitter=jitter, u_fine=u_fine, return_debug=True)
    ss = out['shape_synthesized']
    assert ss['coarse_coords'].shape == (V, M, S, 3) and ss['fine_coords'].shape == (V, M * NI, 3)
    cd = ss['coarse_densities'].cpu().reshape(-1)
    cd_ref = torch.from_numpy(g['coarse_densities'].astype(np.float32)).reshape(-1)
    inb = cd_ref > -1e30
    assert torch.equal(cd > -1e30, inb)
    assert rel_l2(cd[inb], cd_ref[inb]) < 2e-3                      # fp16-stored golden
    assert rel_l2(ss['fine_depths'].cpu().reshape(-1), torch.from_numpy(g['fine_depths'].astype(np.float32)).reshape(-1)) < 2e-3
    for key in ('image_raw', 'image_depth', 'weights_samples', 'image_mask'):
        e = rel_l2(out[key].cpu(), g[key])
        print(tag, key, e)
        assert e < 2e-3, (key, e)


def test_shapenet_renderer_with_the_32_channel_decoder_vs_reference_golden(hip_lib):
    """The ShapeNet launchers' renderer: numeric ray limits, no bbox filter, --decoder_output_dim 32 with --sr_training False (no SR
    module): image_raw = the first 3 of the reference's 32 composited channels (nsr/triplane.py:683)."""
    from oracle import render as orender
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.synth import synth_input, synth_state_dict
    g = golden('render_preset_shapenet64_d32')
    res = int(g['res'])
    tp = Triplane(img_resolution=res, rendering_kwargs=dict(orender.SHAPENET_OPTS), decoder_output_dim=32)
    sd = synth_state_dict({'net.0.weight': (64, 32), 'net.0.bias': (64,), 'net.2.weight': (33, 64), 'net.2.bias': (33,)}, 0)
    sd['net.2.bias'] = sd['net.2.bias'].clone()
    sd['net.2.bias'][0] += float(g['sigma_bias'])
    tp.decoder.load_state_dict(sd, strict=True)
    tp = tp.cuda()
    cams = torch.from_numpy(g['cams'])
    V, M = cams.shape[0], res * res
    planes = synth_input('planes', (V, 96, 128, 128), 3, float(g['plane_scale']))
    gen = torch.Generator().manual_seed(int(g['jitter_seed']))
    jitter = torch.rand(V, M, 64, 1, generator=gen).reshape(V, M, 64)
    u_fine = torch.rand(V * M, 64, generator=gen)
    out = tp(planes.cuda(), cams.cuda(), jitter=jitter, u_fine=u_fine)
    for key in ('image_raw', 'image_depth', 'weights_samples', 'image_mask'):
        e = rel_l2(out[key].cpu(), g[key])
        print('shapenet d32', key, e)
        assert e < 2e-3, (key, e)


@pytest.mark.parametrize("tag", ['objv64_meta', 'shapenet64', 'objv128'])
def test_importance_renderer_seam_outputs_vs_reference_golden(hip_lib, tag):
    """ImportanceRenderer.forward(planes, decoder, ray_origins, ray_directions, rendering_options, return_meta=True) - the reference's
    positional signature - returns the reference's keys (renderer.py:276-300): visibility, and under return_meta weights / all_coords
    / feature_volume; rays as a LIST (M not a square number) give the same per-ray values."""
    from oracle import render as orender
    g, tp, opts, planes, cams, jitter, u_fine, (V, M, S, NI, res) = _preset_scene(tag)
    ro, rd = (t.cuda() for t in orender.make_rays(cams, res))
    pl = planes.cuda().view(V, 3, 32, 128, 128)
    out = tp.renderer(pl, tp.decoder, ro, rd, tp.rendering_kwargs, True, jitter=jitter, u_fine=u_fine)
    assert out['visibility'].shape == (V, M, 1) and out['weights'].shape == (V, M, S + NI - 1, 1)
    assert out['all_coords'].shape == (V, M, S + NI, 3) and out['feature_volume'].shape == (V, M, S + NI, 3)
    for key, gk in (('visibility', 'visibility'), ('weights', 'weights'), ('all_coords', 'all_coords'), ('feature_volume', 'feature_volume')):
        ref = torch.from_numpy(g[gk].astype(np.float32))
        e = rel_l2(out[key].cpu(), ref)
        print(tag, key, e)
        assert e < 2e-3, (key, e)                                    # fp16-stored goldens for the per-sample tensors
    img = out['feature_samples'].permute(0, 2, 1).reshape(V, 3, res, res)
    assert rel_l2(img.cpu(), g['image_raw']) < 2e-3
    # the merged weights sum to weights_samples, and visibility is what is left of the ray
    assert torch.allclose(out['weights'].sum(2), out['weights_samples'], atol=2e-5)
    assert torch.allclose(out['visibility'], 1 - out['weights_samples'], atol=2e-3)
    # a ray LIST: drop 5 rays -> M - 5 (no res x res image behind it); per-ray reductions are unchanged, the call-wide ones
    # (invalid-ray fix-up, depth clamp range) see fewer rays, so compare what does not depend on them
    keep = torch.arange(M)[5:]
    o2 = tp.renderer(pl, tp.decoder, ro[:, keep].contiguous(), rd[:, keep].contiguous(), tp.rendering_kwargs, False,
                     jitter=jitter.reshape(V, M, S)[:, keep], u_fine=u_fine.reshape(V, M, NI)[:, keep].reshape(-1, NI))
    assert o2['feature_samples'].shape == (V, M - 5, 3) and 'weights' not in o2
    hit = (out['weights_samples'][:, keep] > 1e-3).squeeze(-1)        # rays that cross the box are untouched by the fix-up
    assert hit.any()
    assert torch.allclose(o2['feature_samples'][hit], out['feature_samples'][:, keep][hit], atol=1e-5)
    assert torch.allclose(o2['visibility'][hit], out['visibility'][:, keep][hit], atol=1e-5)


def test_generic_marcher_agrees_with_the_lane_per_sample_kernel(hip_lib):
    """The Objaverse preset through both kernels of csrc/render.hip (return_meta selects the generic one): same images to fp32
    summation-order differences."""
    from oracle import render as orender
    g, tp, opts, planes, cams, jitter, u_fine, (V, M, S, NI, res) = _preset_scene('objv64_meta')
    ro, rd = (t.cuda() for t in orender.make_rays(cams, res))
    pl = planes.cuda().view(V, 3, 32, 128, 128)
    a = tp.renderer(pl, tp.decoder, ro, rd, tp.rendering_kwargs, False, jitter=jitter, u_fine=u_fine)
    b = tp.renderer(pl, tp.decoder, ro, rd, tp.rendering_kwargs, True, jitter=jitter, u_fine=u_fine)
    for k in ('feature_samples', 'depth_samples', 'weights_samples', 'visibility'):
        e = rel_l2(b[k], a[k])
        print('generic vs fast', k, e)
        assert e < 2e-5, (k, e)
    ss_a, ss_b = a['shape_synthesized'], b['shape_synthesized']
    assert torch.allclose(ss_a['coarse_coords'], ss_b['coarse_coords'], atol=1e-6)
    inb = ss_a['coarse_densities'] > -1e30
    assert torch.equal(inb, ss_b['coarse_densities'] > -1e30) and rel_l2(ss_b['coarse_densities'][inb], ss_a['coarse_densities'][inb]) < 1e-5


def test_render_repeatability_sweep_100_scenes(hip_lib):
    """r5: the launch-to-launch differences of the SLP build hit 3 - 7 rays of 262 144, scene-dependent
    (profiles/r4_render_spill.md), so repeatability is swept over 100 random scenes - random tri-planes, camera radius 1.2 - 2.3,
    elevation -30 ... +80 degrees, 1 - 6 views at 64^2 - 192^2, both marcher kernels (every fourth scene takes the generic one through
    return_meta's outputs) - three launches each, identical bits."""
    from ln3diff_amd.nsr.triplane import Triplane
    from ln3diff_amd.synth import orbit_cameras
    from oracle import render as orender
    tp = Triplane(img_resolution=128)
    tp.decoder.load_state_dict(_decoder_sd(4.0))
    tp = tp.cuda()
    rng = np.random.RandomState(7)
    rays = 0
    for scene in range(100):
        V, res = int(rng.randint(1, 7)), int(rng.choice([64, 96, 128, 192]))
        rad, el = float(rng.uniform(1.2, 2.3)), float(rng.uniform(-30.0, 80.0))
        g = torch.Generator(device='cuda').manual_seed(1000 + scene)
        pcl = torch.randn(1, 3, 128, 128, 32, device='cuda', generator=g) * float(rng.uniform(1.0, 5.0))
        cams = orbit_cameras(V, radius=rad, elevation_deg=el).cuda()
        j = torch.rand(V, res * res, 64, device='cuda', generator=g)
        u = torch.rand(V * res * res, 64, device='cuda', generator=g)
        idx = torch.zeros(V, dtype=torch.int32, device='cuda')
        rays += V * res * res
        if scene % 4 == 3:
            ro, rd = (t.cuda() for t in orender.make_rays(cams.cpu(), res))
            f = lambda: tp.renderer(None, tp.decoder, ro, rd, tp.rendering_kwargs, True, jitter=j, u_fine=u, planes_channel_last=pcl, plane_index=idx)
            keys = ('feature_samples', 'depth_samples', 'weights_samples', 'visibility', 'weights')
        else:
            f = lambda: tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
            keys = ('image_raw', 'image_depth', 'weights_samples')
        ref = f()
        for _ in range(2):
            o = f()
            for k in keys:
                assert torch.equal(ref[k], o[k]), (scene, V, res, rad, el, k, int((ref[k] != o[k]).sum()))
    print('repeatability sweep: 100 scenes, %.1f M rays x 3 launches, 0 differing values' % (rays / 1e6))


def test_render_with_bit_equal_fine_depths_vs_oracle(hip_lib):
    """r6: the merge's fast path counts the fine samples strictly and detects two bit-equal fine depths by the rank total (such a ray
    takes the full count).  Every ray of this scene carries such a pair (u_fine[:, 7] = u_fine[:, 3]), plus a triple on half of them:
    the images must match the CPU oracle (which sorts) like any other scene, and repeat bitwise."""
    from oracle import render as orender
    from ln3diff_amd.nsr.triplane import Triplane, draw_render_noise
    from ln3diff_amd.synth import synth_input, orbit_cameras
    res, V = 24, 2
    tp = Triplane(img_resolution=res)
    sd = _decoder_sd(4.0)
    tp.decoder.load_state_dict(sd)
    tp = tp.cuda()
    planes = synth_input('planes', (V, 96, 128, 128), 3, 4.0)
    cams = orbit_cameras(V)
    gen = torch.Generator().manual_seed(11)
    jitter, u_fine = draw_render_noise(V, res * res, 64, generator=gen)
    u_fine = u_fine.clone()
    u_fine[:, 7] = u_fine[:, 3]
    u_fine[::2, 40] = u_fine[::2, 3]
    ref = orender.triplane_render(planes, {k: v.float() for k, v in sd.items()}, cams, res, jitter.unsqueeze(-1), u_fine)
    a = tp(planes.cuda(), cams.cuda(), jitter=jitter, u_fine=u_fine)
    b = tp(planes.cuda(), cams.cuda(), jitter=jitter, u_fine=u_fine)
    for key in ('image_raw', 'image_depth', 'weights_samples'):
        assert torch.equal(a[key], b[key]), key
        e = rel_l2(a[key].cpu(), ref[key])
        assert e < 2e-3, (key, e)
