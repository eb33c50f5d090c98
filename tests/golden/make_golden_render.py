"""Golden sections for the renderer and the VAE decode (see make_golden.py).
Runs the reference's Triplane / ImportanceRenderer / OSGDecoder / PatchRaySampler and the
released decoder class from /root/reference (build container only)."""
import io
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

from ln3diff_amd.synth import synth_state_dict, synth_input, orbit_cameras  # noqa: E402
from oracle import render as orender, decoder as odec  # noqa: E402


def _mg():
    import make_golden as mg
    return sys.modules.get('__main__') if hasattr(sys.modules.get('__main__'), 'check') else mg


def ref_rendering_kwargs(patch_res=45):
    opts = dict(orender.OBJAVERSE_OPTS)
    opts.update(image_resolution=256, c_gen_conditioning_zero=True, c_scale=1,
                superresolution_noise_mode='none', density_reg=0.25, density_reg_p_dist=0.004,
                reg_type='l1', decoder_lr_mul=1, decoder_activation='sigmoid', sr_antialias=True,
                return_triplane_features=False, return_sampling_details_flag=True,
                radius_range=[1.5, 2], PatchRaySampler=True, patch_rendering_resolution=patch_res,
                z_near=1.05, z_far=2.45)
    return opts


def build_triplane(res):
    from nsr.triplane import Triplane
    tp = Triplane(25, res, 3, rendering_kwargs=ref_rendering_kwargs(), out_chans=96,
                  triplane_size=224, decoder_in_chans=32, decoder_output_dim=3, sr_kwargs={},
                  bcg_synthesis_kwargs={}, lrm_decoder=False)
    return tp.eval()


def dense_decoder_sd(seed=0):
    """OSGDecoder weights: N(0,1) like the reference init, plus a positive sigma bias so the
    synthetic volume is not empty."""
    shapes = {'net.0.weight': (64, 32), 'net.0.bias': (64,), 'net.2.weight': (4, 64), 'net.2.bias': (4,)}
    sd = synth_state_dict(shapes, seed)
    sd['net.2.bias'] = sd['net.2.bias'].clone()
    sd['net.2.bias'][0] += 4.0
    return sd


def sec_render():
    mg = _mg()
    check, save = mg.check, mg.save
    print('== renderer (reference Triplane.forward vs oracle.render.triplane_render)')
    import torch.nn.functional as F
    # grid_sample restatement
    pl = synth_input('pl', (8, 16, 16), 0)
    g = synth_input('g', (500, 2), 0, 0.7)
    ref = F.grid_sample(pl[None], g[None, None], mode='bilinear', padding_mode='zeros',
                        align_corners=False)[0, :, 0].t()
    check('bilinear_zeros vs F.grid_sample', orender.bilinear_zeros(pl, g[:, 0], g[:, 1]), ref, 1e-6)

    for tag, res, V, plane_scale, sigma_bias in (('dense_r16', 16, 2, 4.0, 4.0), ('dense_r32', 32, 2, 4.0, 4.0),
                                                 ('sparse_r16', 16, 1, 1.0, 0.0)):
        tp = build_triplane(res)
        sd = dense_decoder_sd(0)
        if sigma_bias == 0.0:
            sd['net.2.bias'][0] -= 4.0
        tp.decoder.load_state_dict(sd, strict=True)
        planes = synth_input('planes', (V, 96, 128, 128), 3, plane_scale)
        cams = orbit_cameras(8)[[1, 6][:V]]
        M = res * res
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            r_ref = tp(planes, cams)
        # the reference's RNG stream -> logical tensors
        torch.manual_seed(0)
        jitter = torch.rand(64, V, M, 1).permute(1, 2, 0, 3).contiguous()
        u_fine = torch.rand(V * M, 64)
        o_ref, d_ref, _ = tp.ray_sampler(cams[:, :16].reshape(-1, 4, 4), cams[:, 16:25].reshape(-1, 3, 3), res, res)
        o, d = orender.make_rays(cams, res)
        check(f'{tag} ray origins', o, o_ref, 1e-6)
        check(f'{tag} ray dirs', d, d_ref, 1e-6)
        r = orender.triplane_render(planes, sd, cams, res, jitter, u_fine)
        ss = r_ref['shape_synthesized']
        check(f'{tag} coarse densities', r['detail']['coarse_densities'], ss['coarse_densities'], 1e-4)
        check(f'{tag} fine densities', r['detail']['fine_densities'], ss['fine_densities'], 1e-4)
        check(f'{tag} image_raw', r['image_raw'], r_ref['image_raw'], 1e-4)
        check(f'{tag} image_depth', r['image_depth'], r_ref['image_depth'], 1e-4)
        check(f'{tag} weights_samples', r['weights_samples'], r_ref['weights_samples'], 1e-4)
        print(f'   mask mean {float(r_ref["image_mask"].mean()):.3f} rgb range '
              f'[{float(r_ref["image_raw"].min()):.3f},{float(r_ref["image_raw"].max()):.3f}] '
              f'depth range [{float(r_ref["image_depth"].min()):.4f},{float(r_ref["image_depth"].max()):.4f}]')
        save(f'render_{tag}', image_raw=r_ref['image_raw'], image_depth=r_ref['image_depth'],
             weights_samples=r_ref['weights_samples'], image_mask=r_ref['image_mask'],
             coarse_densities=ss['coarse_densities'].half(), fine_depths=r['detail']['fine_depths'].half(),
             cams=cams, jitter_seed=np.array(0), plane_scale=np.array(plane_scale),
             sigma_bias=np.array(sigma_bias))

    # sigma / rgb grid (triplane_decode_grid -> forward_points -> _run_model, no bbox filter)
    tp = build_triplane(16)
    sd = dense_decoder_sd(0)
    tp.decoder.load_state_dict(sd, strict=True)
    planes = synth_input('planes', (1, 96, 128, 128), 3, 4.0)
    G = 16
    ax = torch.linspace(-0.45, 0.45, G)
    pts = torch.stack(torch.meshgrid(ax, ax, ax, indexing='ij'), dim=-1).reshape(1, -1, 3)
    out = tp.renderer._run_model(planes=planes.reshape(1, 3, 32, 128, 128), decoder=tp.decoder,
                                 sample_coordinates=pts, sample_directions=torch.zeros_like(pts),
                                 options=tp.rendering_kwargs)
    g = orender.decode_grid(planes, sd, G)
    check('grid16 sigma', g['sigma'].reshape(1, -1, 1), out['sigma'], 1e-5)
    check('grid16 rgb', g['rgb'].reshape(1, -1, 3), out['rgb'], 1e-5)
    save('grid16', sigma=out['sigma'].reshape(G, G, G), rgb=out['rgb'].reshape(G, G, G, 3))


PRESETS = {  # tag -> (oracle options, res, V, cameras' orbit radius): the sampling presets of nsr/script_util.py the entry points select
    'shapenet64': (orender.SHAPENET_OPTS, 16, 2, 1.2),
    'objv128': (orender.OBJAVERSE_128_OPTS, 16, 2, 1.7719),
    'objv96': (orender.OBJAVERSE_96_OPTS, 12, 1, 1.7719),
    'eg3d80': (orender.EG3D_80_OPTS, 12, 2, 1.2),
    'afhq48': (orender.AFHQ_48_OPTS, 12, 1, 2.7),
}


def ref_preset_kwargs(opts):
    """rendering_options_defaults (nsr/script_util.py:433-465) + the preset's keys; return_sampling_details_flag to get the per-sample
    tensors and, through it, return_meta (nsr/triplane.py:575-576)."""
    rk = dict(image_resolution=256, disparity_space_sampling=False, clamp_mode='softplus', c_gen_conditioning_zero=True, c_scale=1,
              superresolution_noise_mode='none', density_reg=0.25, density_reg_p_dist=0.004, reg_type='l1', decoder_lr_mul=1,
              decoder_activation='sigmoid', sr_antialias=True, return_triplane_features=False, return_sampling_details_flag=True)
    rk.update(opts)
    return rk


def sec_render_presets():
    """The reference's Triplane.forward under the other rendering presets (numeric ray limits, 48 - 128 samples, no bbox filter,
    black background) and the seam outputs of ImportanceRenderer.forward(return_meta=True): visibility, weights, all_coords,
    feature_volume - against oracle.render, saved as fixtures."""
    mg = _mg()
    check, save = mg.check, mg.save
    print('== renderer presets + return_meta (reference Triplane.forward vs oracle.render)')
    from nsr.triplane import Triplane
    cases = dict(PRESETS, objv64_meta=(orender.OBJAVERSE_OPTS, 16, 2, 1.7719), shapenet64_d32=(orender.SHAPENET_OPTS, 16, 2, 1.2))
    for tag, (opts, res, V, radius) in cases.items():
        if tag == 'shapenet64_d32':
            _shapenet_d32(tag, opts, res, V, radius, check, save)
            continue
        rk = ref_preset_kwargs(opts)
        if 'auto' in str(opts['ray_start']):
            rk.update(PatchRaySampler=True, patch_rendering_resolution=45)       # the Objaverse presets carry it (:788)
        tp = Triplane(25, res, 3, rendering_kwargs=rk, out_chans=96, triplane_size=224, decoder_in_chans=32, decoder_output_dim=3,
                      sr_kwargs={}, bcg_synthesis_kwargs={}, lrm_decoder=False).eval()
        sd = dense_decoder_sd(0)
        tp.decoder.load_state_dict(sd, strict=True)
        planes = synth_input('planes', (V, 96, 128, 128), 3, 4.0)
        cams = orbit_cameras(8, radius=radius)[[1, 6][:V]]
        M, S, NI = res * res, opts['depth_resolution'], opts['depth_resolution_importance']
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            r_ref = tp(planes, cams)
        # the reference's RNG stream -> logical tensors: the 'auto' branch draws on a [S,V,M,1]-strided tensor,
        # the numeric branch on the contiguous [V,M,S,1] repeat (renderer.py:466-471)
        torch.manual_seed(0)
        if opts['ray_start'] == 'auto':
            jitter = torch.rand(S, V, M, 1).permute(1, 2, 0, 3).contiguous()
        else:
            jitter = torch.rand(V, M, S, 1)
        u_fine = torch.rand(V * M, NI)
        o_ref, d_ref, _ = tp.ray_sampler(cams[:, :16].reshape(-1, 4, 4), cams[:, 16:25].reshape(-1, 3, 3), res, res)
        o, d = orender.make_rays(cams, res)
        check(f'{tag} ray origins', o, o_ref, 1e-6)
        check(f'{tag} ray dirs', d, d_ref, 1e-6)
        r = orender.triplane_render(planes, sd, cams, res, jitter, u_fine, opts)
        det = r['detail']
        ss = r_ref['shape_synthesized']
        check(f'{tag} coarse densities', det['coarse_densities'], ss['coarse_densities'], 1e-4)
        check(f'{tag} fine densities', det['fine_densities'], ss['fine_densities'], 1e-4)
        check(f'{tag} coarse coords', det['coarse_coords'], ss['coarse_coords'], 1e-5)
        check(f'{tag} image_raw', r['image_raw'], r_ref['image_raw'], 1e-4)
        check(f'{tag} image_depth', r['image_depth'], r_ref['image_depth'], 1e-4)
        check(f'{tag} weights_samples', r['weights_samples'], r_ref['weights_samples'], 1e-4)
        check(f'{tag} weights', det['weights'], r_ref['weights'], 1e-4)
        check(f'{tag} all_coords', det['all_coords'], r_ref['all_coords'], 1e-5)
        check(f'{tag} feature_volume', det['feature_volume'], r_ref['feature_volume'], 1e-4)
        # 'visibility' only leaves ImportanceRenderer.forward (renderer.py:281): call the seam itself on the same RNG stream
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            r_seam = tp.renderer(planes.reshape(V, 3, 32, 128, 128), tp.decoder, o_ref, d_ref, tp.rendering_kwargs, return_meta=True)
        check(f'{tag} seam feature_samples == Triplane.forward', r_seam['feature_samples'].permute(0, 2, 1).reshape(V, 3, res, res), r_ref['image_raw'], 1e-7)
        check(f'{tag} visibility', det['visibility'], r_seam['visibility'], 1e-4)
        print(f'   {tag}: S {S} NI {NI} mask mean {float(r_ref["image_mask"].mean()):.3f} depth range '
              f'[{float(r_ref["image_depth"].min()):.4f},{float(r_ref["image_depth"].max()):.4f}] visibility mean {float(r_seam["visibility"].mean()):.3f}')
        save(f'render_preset_{tag}', image_raw=r_ref['image_raw'], image_depth=r_ref['image_depth'], weights_samples=r_ref['weights_samples'],
             image_mask=r_ref['image_mask'], visibility=r_seam['visibility'], weights=r_ref['weights'].half(),
             all_coords=r_ref['all_coords'].half(), feature_volume=r_ref['feature_volume'].half(),
             coarse_densities=ss['coarse_densities'].half(), fine_depths=det['fine_depths'].half(), cams=cams, jitter_seed=np.array(0),
             plane_scale=np.array(4.0), sigma_bias=np.array(4.0), res=np.array(res), radius=np.array(radius))


def _shapenet_d32(tag, opts, res, V, radius, check, save):
    """The ShapeNet launchers' renderer: --decoder_output_dim 32 (33-row decoder), --sr_training False -> no SR module; image_raw is the first 3
    of the 32 composited feature channels (nsr/triplane.py:683)."""
    from nsr.triplane import Triplane
    from ln3diff_amd.synth import synth_state_dict
    tp = Triplane(25, res, 3, rendering_kwargs=ref_preset_kwargs(opts), out_chans=96, triplane_size=224, decoder_in_chans=32, decoder_output_dim=32,
                  sr_kwargs={}, bcg_synthesis_kwargs={}, lrm_decoder=False).eval()
    assert tp.superresolution is None
    sd = synth_state_dict({'net.0.weight': (64, 32), 'net.0.bias': (64,), 'net.2.weight': (33, 64), 'net.2.bias': (33,)}, 0)
    sd['net.2.bias'] = sd['net.2.bias'].clone()
    sd['net.2.bias'][0] += 4.0
    tp.decoder.load_state_dict(sd, strict=True)
    planes = synth_input('planes', (V, 96, 128, 128), 3, 4.0)
    cams = orbit_cameras(8, radius=radius)[[1, 6][:V]]
    M, S, NI = res * res, opts['depth_resolution'], opts['depth_resolution_importance']
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        r_ref = tp(planes, cams)
    torch.manual_seed(0)
    jitter = torch.rand(V, M, S, 1)
    u_fine = torch.rand(V * M, NI)
    sd4 = dict(sd)
    sd4['net.2.weight'], sd4['net.2.bias'] = sd['net.2.weight'][:4], sd['net.2.bias'][:4]
    r = orender.triplane_render(planes, sd4, cams, res, jitter, u_fine, opts)
    assert r_ref['feature_image'].shape[1] == 32
    check(f'{tag} image_raw (= feature_image[:, :3])', r['image_raw'], r_ref['image_raw'], 1e-4)
    check(f'{tag} image_depth', r['image_depth'], r_ref['image_depth'], 1e-4)
    check(f'{tag} weights_samples', r['weights_samples'], r_ref['weights_samples'], 1e-4)
    save(f'render_preset_{tag}', image_raw=r_ref['image_raw'], image_depth=r_ref['image_depth'], weights_samples=r_ref['weights_samples'],
         image_mask=r_ref['image_mask'], cams=cams, jitter_seed=np.array(0), plane_scale=np.array(4.0), sigma_bias=np.array(4.0),
         res=np.array(res), radius=np.array(radius))


# ------------------------------------------------------------------ VAE decode
def build_decoder(hidden, depth, heads):
    """The released decoder class (vit/vit_triplane.py:1982) around a DiT2 of the given size."""
    from dit.dit_decoder import DiT2
    from vit import vit_triplane as vt
    with contextlib.redirect_stdout(io.StringIO()):
        vit_decoder = DiT2(input_size=16, patch_size=2, in_channels=hidden, hidden_size=hidden, depth=depth,
                           num_heads=heads, num_classes=0, learn_sigma=False, mixed_prediction=False,
                           context_dim=None, roll_out=True, plane_n=3, return_all_layers=False)
        tp = build_triplane(128)
        cls = vt.RodinSR_256_fusionv6_ConvQuant_liteSR_dinoInit3DAttn_SD_B_3L_C_withrollout_withSD_D_ditDecoder
        dec = cls(vit_decoder=vit_decoder, triplane_decoder=tp, cls_token=False, sr_ratio=2, vae_p=2,
                  ldm_z_channels=4, ldm_embed_dim=4)
    return dec.eval()


def sec_decoder():
    mg = _mg()
    check, save, load_synth, manifest_json = mg.check, mg.save, mg.load_synth, mg.manifest_json
    print('== VAE decode (reference released decoder class vs oracle.decoder.vae_decode)')
    for tag, (hidden, depth, heads), B in (('tiny', (128, 2, 2), 2), ('dit2_l2', (1024, 24, 16), 1)):
        dec = build_decoder(hidden, depth, heads)
        pe_ref = dec.vit_decoder.pos_embed.clone()
        check(f'{tag} decoder pos_embed', odec.decoder_pos_embed(hidden), pe_ref, 1e-6)
        sd, shapes = load_synth(dec, 0)
        latent = synth_input('latent', (B, 12, 32, 32), 5)
        with contextlib.redirect_stdout(io.StringIO()):
            tok_ref = dec.vit_decode_backbone({'latent_normalized_2Ddiffusion': latent}, 128)
            ret = dec.vit_decode_postprocess(tok_ref, {})
        planes_ref = ret['latent_after_vit']
        tok = odec.vae_decode(sd, latent, heads, return_tokens=True)
        check(f'{tag} DiT2 tokens', tok, tok_ref, 5e-5)
        planes = odec.vae_decode(sd, latent, heads)
        check(f'{tag} planes', planes, planes_ref, 5e-5)
        print(f'   planes std {float(planes_ref.std()):.3f}')
        keep = {k: v for k, v in shapes.items()
                if k.startswith(('vit_decoder.', 'superresolution.ldm_upsample', 'superresolution.conv_sr',
                                 'triplane_decoder.decoder'))}
        save(f'decode_{tag}', planes_sub=planes_ref[:, :, ::8, ::8], tokens_sub=tok_ref[:, ::16, ::8],
             planes_mean=planes_ref.mean(), planes_std=planes_ref.std(), manifest=manifest_json(keep),
             all_keys=manifest_json(shapes))
        del dec, sd
