"""Sampling drivers on the HIP path - restatements of the reference's engine methods:

  DiffusionEngineLSGM.sample / eval_cldm      nsr/lsgm/sgm_DiffusionEngine.py:386-480   -> T23DPipeline
  FlowMatchingEngine.sample / eval_cldm /
      eval_i23d_and_export                     nsr/lsgm/flow_matching_trainer.py:510-760 -> FlowMatchingEngine (alias I23DPipeline)
  TrainLoop.render_video_given_triplane        nsr/train_util_diffusion.py:177-300       -> render_video_given_triplane

What is kept: seeding (th.manual_seed(41) / seed 42 before the z draw), z shape, `repeat_interleave` of one condition over
num_samples, the unconditional branch = ZERO embeddings (`force_uc_zero_embeddings`), the CFG concat order (sgm: [uc, c];
flow matching: [c, uc]), `planes *= triplane_scaling_divider`, the AE behaviour strings, 40 cameras for the T23D video and
`camera[:24]` for the I23D one, marching at threshold 10 on a 192^3 grid scaled to +-0.45 and rotated -90 deg about x.
What is dropped: video encoding / logging, `th.cuda.empty_cache()` calls, and the one-camera-per-call loop: all views of all
samples are one renderer launch with `views_per_call=1`, which keeps the call-wide reductions of the reference's renderer (depth
clamp range, ray-limit fix-up) per camera exactly as the loop has them.
"""
import torch

from .sgm.sampling import DiscreteDenoiser, EulerEDMSampler, VanillaCFG

TRIPLANE_SCALING_DIVIDER = 0.96806      # the released Objaverse runs (shell_scripts/final_release/inference/*.sh); a CLI flag there


@torch.no_grad()
def render_video_given_triplane(planes, rec_model, cams, triplane_scaling_divider=TRIPLANE_SCALING_DIVIDER, latent_name='latent_normalized_2Ddiffusion',
                                export_mesh=False, mesh_size=192, mesh_thres=10, mesh_path=None, resolution=None, jitter=None, u_fine=None):
    """planes: sampled latent [B, 12, 32, 32] (scaled IN PLACE like the reference, :188); rec_model: `AE`; cams [V, 25] rendered
    for every sample.  Returns {'latent_after_vit' (if produced), 'image_raw' [B,V,3,R,R], 'image_depth', 'weights_samples',
    'image_mask', 'mesh': [(verts, faces, colors)] when export_mesh}."""
    planes *= triplane_scaling_divider
    ddpm_latent = {latent_name: planes}
    ddpm_latent.update(rec_model(latent=ddpm_latent, behaviour='decode_after_vae_no_render'))
    out = {}
    if export_mesh:
        from .mesh import mesh_from_grid
        grid_out = rec_model(latent=ddpm_latent, grid_size=mesh_size, behaviour='triplane_decode_grid')
        out['mesh'] = [mesh_from_grid(rec_model.decoder, ddpm_latent, grid_out['sigma'][i], mesh_size, mesh_thres, sample_index=i,
                                      path=(mesh_path.format(i) if mesh_path else None)) for i in range(planes.shape[0])]
    B, V = planes.shape[0], cams.shape[0]
    kw = {}
    if resolution is not None:
        kw['neural_rendering_resolution'] = resolution
    pcl = ddpm_latent.get('planes_channel_last')
    if pcl is not None:
        kw['plane_index'] = torch.arange(B, device=planes.device, dtype=torch.int32).repeat_interleave(V)
    # one camera per reference call (:262-283): call-wide reductions of the renderer are per view
    pred = rec_model(img=None, c=cams.repeat(B, 1), latent=ddpm_latent, behaviour='triplane_dec', jitter=jitter, u_fine=u_fine, views_per_call=1, **kw)
    for k in ('image_raw', 'image_depth', 'weights_samples', 'image_mask'):
        out[k] = pred[k].view(B, V, *pred[k].shape[1:])
    if 'latent_after_vit' in ddpm_latent:
        out['latent_after_vit'] = ddpm_latent['latent_after_vit']
    out['planes_channel_last'] = pcl
    return out


@torch.no_grad()
def render_pairs(latent_all, rec_model, cams, pairs, triplane_scaling_divider=TRIPLANE_SCALING_DIVIDER, resolution=None, noise_seed=None,
                 latent_name='latent_normalized_2Ddiffusion'):
    """The (sample, view) units of one rank (parallel.shard_pairs): decode the samples that occur in `pairs` ONCE, render all
    listed views in one launch (views_per_call=1: the reference's one-camera-per-call reductions, so a view's pixels do not depend
    on which other views share the launch).  latent_all [B, 12, 32, 32] is NOT modified.  noise_seed: None draws the stratified /
    importance noise from the device's default generator in bulk; an int seeds a generator per (sample, view) pair, which makes a
    frame independent of the world size bit for bit.  Returns {'image_raw' [P,3,R,R], 'image_depth', 'weights_samples',
    'image_mask', 'pair_index' [P, 2] = (sample, view)}."""
    from .nsr.triplane import draw_render_noise
    dev = latent_all.device
    res = resolution or rec_model.decoder.triplane_decoder.neural_rendering_resolution
    V_all = cams.shape[0]
    if not pairs:
        z = lambda c: torch.empty(0, c, res, res, device=dev)
        return {'image_raw': z(3), 'image_depth': z(1), 'weights_samples': z(1), 'image_mask': z(1),
                'pair_index': torch.empty(0, 2, dtype=torch.int64, device=dev)}
    samples = sorted({s for s, _, _ in pairs})
    local = {s: i for i, s in enumerate(samples)}
    planes = latent_all[samples].clone()
    planes *= triplane_scaling_divider
    ddpm_latent = {latent_name: planes}
    ddpm_latent.update(rec_model(latent=ddpm_latent, behaviour='decode_after_vae_no_render'))
    c = torch.cat([cams[v0:v1] for _, v0, v1 in pairs])
    pidx = torch.cat([torch.full((v1 - v0,), local[s], dtype=torch.int32, device=dev) for s, v0, v1 in pairs])
    pair_index = torch.tensor([(s, v) for s, v0, v1 in pairs for v in range(v0, v1)], dtype=torch.int64, device=dev)
    jitter = u_fine = None
    if noise_seed is not None:
        rk_ = rec_model.decoder.triplane_decoder.rendering_kwargs
        js, us = [], []
        for s, v in pair_index.tolist():
            g = torch.Generator(device=dev).manual_seed(int(noise_seed) + s * V_all + v)
            j, u = draw_render_noise(1, res * res, rk_.get('depth_resolution', 64), generator=g, device=dev,
                                     n_importance=rk_.get('depth_resolution_importance', 64))
            js.append(j)
            us.append(u)
        jitter, u_fine = torch.cat(js), torch.cat(us)
    pred = rec_model(img=None, c=c, latent=ddpm_latent, behaviour='triplane_dec', jitter=jitter, u_fine=u_fine, views_per_call=1,
                     plane_index=pidx, neural_rendering_resolution=res)
    out = {k: pred[k] for k in ('image_raw', 'image_depth', 'weights_samples', 'image_mask')}
    out['pair_index'] = pair_index
    return out


def _zero_uc(cond):
    """GeneralConditioner.get_unconditional_conditioning(..., force_uc_zero_embeddings=[cond_key]) (encoders/modules.py:161-163,
    sgm_DiffusionEngine.py:448-452): every tensor of the unconditional branch is zeros."""
    return {k: torch.zeros_like(v) for k, v in cond.items()}


class T23DPipeline:
    """Text -> 3D: EulerEDM (LegacyDDPM sigmas) + VanillaCFG over DiT_TriLatent, then decode + render."""

    def __init__(self, dit, decoder, num_steps=250, cfg_scale=6.5, conditioner=None, triplane_scaling_divider=TRIPLANE_SCALING_DIVIDER,
                 img_size=128):
        from .nsr.script_util import AE
        self.dit, self.decoder, self.conditioner = dit, decoder, conditioner
        self.rec_model = decoder if isinstance(decoder, AE) else AE(None, decoder, img_size)
        self.sampler = EulerEDMSampler(num_steps=num_steps, guider=VanillaCFG(cfg_scale))
        self.denoiser = DiscreteDenoiser()
        self.triplane_scaling_divider = triplane_scaling_divider

    @torch.no_grad()
    def encode_prompts(self, captions, uc_captions=None):
        """GeneralConditioner semantics for the T23D config (sgm/modules/encoders/modules.py:80-191): one FrozenCLIPEmbedder on key
        'caption' -> cond = {'crossattn': last_hidden_state, 'vector': pooled}.  The unconditional branch is ZEROS (what the
        sampling driver asks for with force_uc_zero_embeddings, sgm_DiffusionEngine.py:448-452); pass uc_captions only to get the
        legacy "encode the empty prompt" behaviour.  captions: list of strings or int token ids [B, 77]."""
        assert self.conditioner is not None, "construct the pipeline with conditioner=FrozenCLIPEmbedder(...)"
        z, pooled = self.conditioner(captions)
        cond = {'crossattn': z, 'vector': pooled}
        if uc_captions is None:
            return cond, _zero_uc(cond)
        zu, pu = self.conditioner(uc_captions)
        return cond, {'crossattn': zu.clone(), 'vector': pu.clone()}

    @torch.no_grad()
    def sample_latent(self, z, cond, uc=None):
        # DiffusionEngineLSGM.sample (:401-404): the sampler gets the closure `(input, sigma, c) -> denoised` over denoiser + model
        return self.sampler(self.denoiser.bind(self.dit), z, cond, uc=_zero_uc(cond) if uc is None else uc)

    # DiffusionEngineLSGM.sample (:386-407): z ~ randn(seed), shape [N, 3*C, S, S]
    @torch.no_grad()
    def sample(self, cond, uc=None, batch_size=1, shape=None, seed=41, device=None):
        device = device or cond['crossattn'].device
        shape = shape or (3 * self.dit.in_channels if self.dit.roll_out else self.dit.in_channels, 32, 32)
        torch.manual_seed(seed)
        z = torch.randn(batch_size, *shape).to(device)
        return self.sample_latent(z, cond, uc)

    # DiffusionEngineLSGM.eval_cldm (:410-480): one condition x num_samples, then the video of every sample
    @torch.no_grad()
    def eval_cldm(self, cond, cams, num_samples=1, resolution=None, export_mesh=False, seed=41, **render_kw):
        assert cond['crossattn'].shape[0] == 1
        c = {k: v.repeat_interleave(num_samples, 0) for k, v in cond.items()}
        latent = self.sample(c, None, batch_size=num_samples, seed=seed)
        return latent, self.render_video_given_triplane(latent.clone(), cams[:40], resolution=resolution, export_mesh=export_mesh, **render_kw)

    @torch.no_grad()
    def render_video_given_triplane(self, planes, cams, **kw):
        return render_video_given_triplane(planes, self.rec_model, cams, self.triplane_scaling_divider, **kw)

    @torch.no_grad()
    def decode(self, latent, want_nchw=False):
        lat = latent * self.triplane_scaling_divider
        dec = self.rec_model.decoder
        tok = dec.vit_decode_backbone({'latent_normalized_2Ddiffusion': lat}, 128)
        return dec.vit_decode_postprocess(tok, {}, want_nchw=want_nchw)

    @torch.no_grad()
    def render(self, dec_out, cams, res, jitter=None, u_fine=None):
        """cams [V,25] rendered for EVERY sample: returns images [B, V, 3, res, res]."""
        pcl = dec_out['planes_channel_last']
        B, V = pcl.shape[0], cams.shape[0]
        c = cams.repeat(B, 1)
        idx = torch.arange(B, device=pcl.device, dtype=torch.int32).repeat_interleave(V)
        out = self.rec_model.decoder.triplane_decoder(c=c, planes_channel_last=pcl, plane_index=idx,
                                                      neural_rendering_resolution=res, jitter=jitter, u_fine=u_fine, views_per_call=1)
        return {k: (v.view(B, V, *v.shape[1:]) if torch.is_tensor(v) else v) for k, v in out.items()
                if k in ('image_raw', 'image_depth', 'weights_samples', 'image_mask')}

    @torch.no_grad()
    def __call__(self, z, cond, uc, cams, res):
        latent = self.sample_latent(z, cond, uc)
        dec = self.decode(latent)
        img = self.render(dec, cams, res)
        return latent, img


class FlowMatchingEngine:
    """Image -> 3D (and the flow-matching T23D variant): transport ODE sampler over `ddpm_model.forward_with_cfg`.
    `sampling_method` defaults to 'dopri5' like the reference's `sample_ode(num_steps=num_steps, cfg=True)` (transport/transport.py:377:
    torchdiffeq dopri5, atol 1e-6, rtol 1e-3, output at linspace(0, 1, num_steps)[-1]; restated with torchdiffeq 0.2.3's step
    controller and dense output in transport/__init__.py - torchdiffeq itself is not installed here).  The benchmark
    configurations (BASELINE configs[2] / [4]: "50 steps") use the fixed-step 'euler' integrator and say so explicitly."""

    def __init__(self, ddpm_model, decoder, conditioner=None, triplane_scaling_divider=TRIPLANE_SCALING_DIVIDER, img_size=128,
                 path_type='Linear', prediction='velocity', snr_type='lognorm', sampling_method='dopri5'):
        from .nsr.script_util import AE
        from .transport import Sampler, create_transport
        self.ddpm_model, self.conditioner = ddpm_model, conditioner
        self.rec_model = decoder if isinstance(decoder, AE) else AE(None, decoder, img_size)
        self.transport = create_transport(path_type=path_type, prediction=prediction, snr_type=snr_type)
        self.transport_sampler = Sampler(self.transport)
        self.sampling_method = sampling_method
        self.triplane_scaling_divider = triplane_scaling_divider

    # FlowMatchingEngine.sample (:510-552)
    @torch.no_grad()
    def sample(self, cond, uc=None, batch_size=16, shape=None, use_cfg=True, cfg_scale=4.0, num_steps=250, seed=42, zs=None, **kwargs):
        assert use_cfg
        uc = _zero_uc(cond) if uc is None else uc
        dev = next(iter(cond.values())).device
        shape = shape or (3 * self.ddpm_model.in_channels if self.ddpm_model.roll_out else self.ddpm_model.in_channels, 32, 32)
        if zs is None:
            torch.manual_seed(seed)
            zs = torch.randn(batch_size, *shape).to(dev)
        c_out = {k: torch.cat((cond[k], uc[k]), 0) for k in cond if k in ('vector', 'crossattn', 'concat')}     # [c, uc]
        zs = torch.cat([zs, zs], 0)
        fn = self.transport_sampler.sample_ode(sampling_method=self.sampling_method, num_steps=num_steps)
        cache = self.ddpm_model.prepare_context(c_out) if hasattr(self.ddpm_model, 'prepare_context') else None
        kw = dict(context_cache=cache) if cache is not None else dict(context=c_out)
        samples = fn(zs, self.ddpm_model.forward_with_cfg, return_trajectory=False, cfg_scale=cfg_scale, **kw)[-1]
        self.last_ode_stats = getattr(fn, 'last_stats', None)        # adaptive solvers: network evaluations, attempted / accepted steps, end time
        return samples.chunk(2, dim=0)[0].contiguous()                                                            # drop the null half

    # FlowMatchingEngine.eval_cldm (:554-682): one condition x num_samples; camera[:24]
    @torch.no_grad()
    def eval_cldm(self, cond, camera, num_samples=1, unconditional_guidance_scale=4.0, num_steps=250, seed=42, export_mesh=False,
                  resolution=None, **render_kw):
        assert cond['crossattn'].shape[0] == 1
        c = {k: v.repeat_interleave(num_samples, 0) for k, v in cond.items()}
        samples = self.sample(c, None, batch_size=num_samples, cfg_scale=unconditional_guidance_scale, num_steps=num_steps, seed=seed)
        return samples, render_video_given_triplane(samples.clone(), self.rec_model, camera[:24], self.triplane_scaling_divider,
                                                    export_mesh=export_mesh, resolution=resolution, **render_kw)

    # FlowMatchingEngine.eval_i23d_and_export (:684-760): image -> conditioner -> sample -> mesh + video
    @torch.no_grad()
    def eval_i23d_and_export(self, inp_img, camera, num_steps=250, seed=42, mesh_size=192, mesh_thres=10, unconditional_guidance_scale=4.0,
                             num_samples=1, export_mesh=True, resolution=None, mesh_path=None):
        assert self.conditioner is not None, "construct the engine with conditioner=I23DConditioner(...)"
        cond = self.conditioner(inp_img)
        samples, out = self.eval_cldm(cond, camera, num_samples, unconditional_guidance_scale, num_steps, seed, export_mesh,
                                      resolution, mesh_size=mesh_size, mesh_thres=mesh_thres, mesh_path=mesh_path)
        return samples, out


I23DPipeline = FlowMatchingEngine


class GuidedDiffusionEngine:
    """The guided_diffusion engines of the ShapeNet / FFHQ entry point (scripts/vit_triplane_diffusion_sample.py):
    TrainLoop3DDiffusion.eval_ddpm_sample (nsr/train_util_diffusion.py:863-921) and TrainLoop3DDiffusionLSGM_crossattn.eval_cldm
    (nsr/lsgm/crossattn_cldm.py:510-640): `SpacedDiffusion.p_sample_loop` - or `ddim_sample_loop` with classifier-free guidance
    when use_ddim - over the denoiser, then render_video_given_triplane per sample.  The reference passes mixing_normal=True,
    which reads `ddpm_model.mixing_logit`; its DiT classes do not define one (dit/dit_models_xformers.py:767-772 is commented
    out), so that branch only ever ran with the U-Net denoiser (ln3diff_amd.guided_diffusion.unet.UNetModel, r5): mixing is applied
    for a denoiser built with mixed_prediction, and is off for the DiTs."""

    def __init__(self, ddpm_model, decoder, diffusion, conditioner=None, triplane_scaling_divider=1.0, img_size=128, batch_size=1,
                 diffusion_input_size=32):
        from .nsr.script_util import AE
        self.ddpm_model, self.diffusion, self.conditioner = ddpm_model, diffusion, conditioner
        self.rec_model = decoder if isinstance(decoder, AE) else AE(None, decoder, img_size)
        self.triplane_scaling_divider = triplane_scaling_divider
        self.batch_size, self.diffusion_input_size = batch_size, diffusion_input_size

    def _noise_size(self, batch_size, overwrite_diff_inp_size=None):
        m = self.ddpm_model
        S = int(overwrite_diff_inp_size) if overwrite_diff_inp_size else self.diffusion_input_size
        return (batch_size, 3 * m.in_channels if m.roll_out else m.in_channels, S, S)

    @torch.no_grad()
    def sample(self, cond=None, batch_size=None, use_ddim=False, unconditional_guidance_scale=1.0, clip_denoised=False, noise=None,
               overwrite_diff_inp_size=None, device=None):
        B = batch_size or self.batch_size
        shape = self._noise_size(B, overwrite_diff_inp_size)
        device = device or next(self.ddpm_model.parameters()).device
        if cond is not None:
            c = cond['crossattn'] if isinstance(cond, dict) and 'crossattn' in cond else cond
            c = c if not torch.is_tensor(c) or c.shape[0] == B else c.repeat_interleave(B, 0)        # broadcast to batch_size (:573-577)
        else:
            c = None
        if noise is None:
            noise = torch.randn(*shape, device=device)
        # mixing_normal=True as the reference passes it (crossattn_cldm.py:543-547, train_util_diffusion.py:888): read by the denoiser
        # that HAS a mixing_logit - the U-Net with mixed_prediction; the DiT classes define none (see the class docstring)
        mix = bool(getattr(self.ddpm_model, 'mixed_prediction', False)) and hasattr(self.ddpm_model, 'mix')
        if use_ddim:
            return self.diffusion.ddim_sample_loop(self.ddpm_model, shape, cond=c, noise=noise, clip_denoised=clip_denoised,
                                                   unconditional_guidance_scale=unconditional_guidance_scale, device=device, mixing_normal=mix)
        return self.diffusion.p_sample_loop(self.ddpm_model, shape, cond=c, noise=noise, clip_denoised=clip_denoised, device=device,
                                            mixing_normal=mix)

    @torch.no_grad()
    def eval_cldm(self, cond, camera, use_ddim=False, unconditional_guidance_scale=1.0, export_mesh=False, resolution=None,
                  overwrite_diff_inp_size=None, **render_kw):
        latent = self.sample(cond, use_ddim=use_ddim, unconditional_guidance_scale=unconditional_guidance_scale,
                             overwrite_diff_inp_size=overwrite_diff_inp_size)
        return latent, render_video_given_triplane(latent.clone(), self.rec_model, camera, self.triplane_scaling_divider,
                                                   export_mesh=export_mesh, resolution=resolution, **render_kw)

    @torch.no_grad()
    def eval_ddpm_sample(self, camera, **kw):
        return self.eval_cldm(None, camera, **kw)
