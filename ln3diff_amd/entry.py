"""Shared body of the two sampling entry points (scripts/vit_triplane_diffusion_sample_objaverse.py and
scripts/vit_triplane_diffusion_sample.py): flag surface of the reference launchers, model construction, conditioning, sharding over
ranks, engine dispatch, outputs.

Reference behaviour kept: flag names (shell_scripts/final_release/inference/*.sh, the two `create_argparser`s), `torchrun` launch
with LOCAL_RANK, th.manual_seed before the noise draw, one condition x --num_samples, `planes *= --triplane_scaling_divider`,
engine per --trainer_name (scripts/..._objaverse.py:135-142), outputs under --logdir.  Differences a user sees: every rank samples
its shard of the batch (the reference samples on rank 0 only, :170); conditioning comes from --cond_path tensors, the HIP CLIP /
DINO conditioners (--clip_checkpoint ...) or is synthesised; weights come from --resume_checkpoint or are synthesised.
"""
import argparse
import json
import os

import numpy as np
import torch

# --trainer_name -> engine (scripts/vit_triplane_diffusion_sample_objaverse.py:135-142, scripts/vit_triplane_diffusion_sample.py:236-242)
TRAINERS = {
    'sgm_legacy': 'edm', 'sgm': 'edm',                                  # DiffusionEngineLSGM: EulerEDM + VanillaCFG
    'flow_matching': 'flow', 'flow_matching_gs': 'flow',                # FlowMatchingEngine
    'ddpm': 'gd', 'adm': 'gd', 'vpsde_crossattn': 'gd', 'vpsde_crossattn_objv': 'gd',      # guided_diffusion p_sample / ddim loops
}
# flags of the reference launchers that only matter for training / data loading: accepted, recorded in args.json, unused
_IGNORED_DEFAULTS = dict(lr=5e-5, batch_size=1, microbatch=-1, ema_rate='0.9999', log_interval=50, eval_interval=2500,
                         save_interval=10000, use_fp16=False, use_amp=False, data_dir='', eval_data_dir='', num_workers=1,
                         use_lmdb=False, use_wds=False, objv_dataset=True, iterations=150000, weight_decay=0.0, lr_anneal_steps=0,
                         schedule_sampler='uniform', anneal_lr=False, load_submodule_name='', ignore_resume_opt=False,
                         freeze_ae=False, denoised_ae=True, overfitting=False, allow_tf32=True, save_img=False,
                         use_train_trajectory=False, cond_key='caption', use_eos_feature=False, interval=1, eval_batch_size=1)


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('boolean value expected')


def add_dict_to_argparser(ap, d):            # guided_diffusion/script_util.py:712-722
    for k, v in d.items():
        t = str if v is None else (str2bool if isinstance(v, bool) else type(v))
        ap.add_argument(f'--{k}', default=v, type=t)


def create_argparser(objaverse=True):
    d = dict(
        dit_model_arch='DiT-L/2' if objaverse else 'DiT-B/2', arch_dit_decoder='DiT2-L/2' if objaverse else 'DiT2-B/2', i23d=False,
        trainer_name='sgm_legacy' if objaverse else 'adm', num_samples=4 if objaverse else 10,
        unconditional_guidance_scale=6.5 if objaverse else 1.0, triplane_scaling_divider=0.96806 if objaverse else 1.0,
        timestep_respacing='250' if objaverse else '', diffusion_steps=1000, noise_schedule='linear', sample_steps=250,
        ode_method='euler', use_ddim=False, clip_denoised=False, image_size=128, num_views=40 if objaverse else 24, export_mesh=False,
        mesh_grid=192, mesh_thres=10.0, logdir='./logs/sample', resume_checkpoint='', ddpm_model_path='', rec_model_path='',
        cond_path='', pose_path='', seed=41 if objaverse else 0, context_dim=768, learn_sigma=False, denoise_in_channels=4,
        diffusion_input_size=32, roll_out=True, prompt='' if objaverse else 'a red chair', cfg='objverse_tuneray_aug_resolution_64_64_auto' if objaverse else 'shapenet',
        mv_input=False, num_mv_views=4, clip_checkpoint='', dino_checkpoint='', tokenizer_dir='', image_path='',
        overwrite_diff_inp_size='', create_controlnet=False)
    d.update(_IGNORED_DEFAULTS)
    ap = argparse.ArgumentParser(allow_abbrev=False)
    add_dict_to_argparser(ap, d)
    return ap


def validate(args):
    """Flag combinations that cannot run are refused up front with the reason."""
    from .dit.dit_trilatent import DiT_models as T23D
    from .dit.dit_i23d import DiT_models as I23D
    from .dit.dit_decoder import DiT2_models
    if args.trainer_name not in TRAINERS:
        raise SystemExit(f"--trainer_name {args.trainer_name}: known engines are {sorted(TRAINERS)}")
    kind = TRAINERS[args.trainer_name]
    if args.create_controlnet or 'cldm' in args.trainer_name:
        raise SystemExit("ControlNet engines are outside the sampling hot path (SURVEY.md section 8)")
    reg = I23D if args.i23d else T23D
    if args.dit_model_arch not in reg:
        other = T23D if args.i23d else I23D
        hint = " (that is an %s architecture: %s --i23d)" % (("T23D", "drop") if args.i23d else ("I23D", "pass")) if args.dit_model_arch in other else ""
        raise SystemExit(f"--dit_model_arch {args.dit_model_arch}: not in the {'I23D' if args.i23d else 'T23D'} registry {sorted(reg)}{hint}; "
                         "U-Net denoisers are outside the hot path")
    if args.arch_dit_decoder not in DiT2_models:
        raise SystemExit(f"--arch_dit_decoder {args.arch_dit_decoder}: known {sorted(DiT2_models)}")
    pixart_t23d = (not args.i23d) and args.dit_model_arch.startswith('DiT-PixelArt')
    if kind == 'flow' and not args.i23d and not pixart_t23d:
        raise SystemExit("--trainer_name flow_matching needs an I23D denoiser (--i23d true) or the PixArt-style T23D one "
                         "(--dit_model_arch DiT-PixelArt-L/2): the flow-matching engine calls forward_with_cfg(x, t, context=..., "
                         "cfg_scale=...), which the plain T23D DiT_TriLatent does not define (in the reference neither: "
                         "dit/dit_models_xformers.py:915 takes class labels)")
    if pixart_t23d and kind != 'flow':
        raise SystemExit("--dit_model_arch DiT-PixelArt-* (DiT_TriLatent_PixelArt) is the flow-matching T23D denoiser: use "
                         "--trainer_name flow_matching")
    if kind == 'gd' and not args.use_ddim and args.unconditional_guidance_scale != 1.0:
        print("[entry] note: p_sample_loop applies no classifier-free guidance (the reference forwards the scale to ddim_sample_loop "
              "only, crossattn_cldm.py:543-547); --unconditional_guidance_scale is ignored without --use_ddim true")
    if args.i23d and kind != 'flow':
        raise SystemExit("--i23d models are flow-matching models: use --trainer_name flow_matching (the released I23D launcher does)")
    if 'MV' in args.dit_model_arch and not args.mv_input:
        args.mv_input = True
    if args.num_samples < 1 or args.num_views < 1:
        raise SystemExit("--num_samples and --num_views must be >= 1")
    return kind


def build_models(args, dev, rank):
    from .dit.dit_trilatent import DiT_models as T23D
    from .dit.dit_models_xformers import TextCondDiTBlock
    from .dit.dit_i23d import DiT_models as I23D
    from .dit.dit_decoder import DiT2_models
    from .nsr.triplane import Triplane
    from .nsr.script_util import AE
    from .vit.vit_triplane import RodinSR_256_fusionv6_ConvQuant_liteSR_dinoInit3DAttn_SD_B_3L_C_withrollout_withSD_D_ditDecoder as Dec
    from .synth import fill_module_random_
    from .checkpoint import load_checkpoint
    common = dict(input_size=args.diffusion_input_size, num_classes=0, learn_sigma=args.learn_sigma,
                  in_channels=args.denoise_in_channels, roll_out=args.roll_out)
    if args.i23d:
        dit = I23D[args.dit_model_arch](context_dim=1024, pooling_ctx_dim=768, **common)
    else:
        dit = T23D[args.dit_model_arch](context_dim=args.context_dim, vit_blk=TextCondDiTBlock, **common)
    vit = DiT2_models[args.arch_dit_decoder](input_size=16, num_classes=0, learn_sigma=False, in_channels=dit.embed_dim,
                                             mixed_prediction=False, context_dim=None, roll_out=True, plane_n=3)
    dec = Dec(vit_decoder=vit, triplane_decoder=Triplane(img_resolution=args.image_size), cls_token=False, vae_p=2, ldm_z_channels=4,
              ldm_embed_dim=4)
    dit, dec = dit.to(dev), dec.to(dev)
    ckpts = [p for p in (args.resume_checkpoint, args.ddpm_model_path, args.rec_model_path) if p]
    if rank == 0:
        if not ckpts:
            fill_module_random_(dit, 0, dev)
            fill_module_random_(dec, 1, dev)
            dec.triplane_decoder.decoder.net[2].bias.data[0] += 4.0
        for p in ckpts:
            load_checkpoint(p, dit=dit if p != args.rec_model_path else None, decoder=dec if p != args.ddpm_model_path else None)
    return dit, AE(None, dec, args.image_size), dec


def load_conditioning(args, dev):
    """{'crossattn', 'vector'[, 'concat']} for P prompts: --cond_path tensors, the HIP conditioners on --prompt / --image_path when
    their checkpoints are given, else synthetic tensors of the right shapes."""
    from .synth import synth_input
    if args.cond_path:
        raw = torch.load(args.cond_path) if args.cond_path.endswith('.pt') else dict(np.load(args.cond_path))
        return {k: torch.as_tensor(v).float() for k, v in raw.items()}
    if args.clip_checkpoint and args.prompt and not args.i23d:
        from .sgm.encoders import FrozenCLIPEmbedder
        from .checkpoint import load_checkpoint
        enc = FrozenCLIPEmbedder(device=str(dev), always_return_pooled=True, tokenizer_dir=args.tokenizer_dir or None)
        load_checkpoint(args.clip_checkpoint, conditioner=enc)
        z, pooled = enc.to(dev)([args.prompt])
        return {'crossattn': z.float().cpu(), 'vector': pooled.float().cpu()}
    if args.i23d:
        mv = args.mv_input                # MVCond: CLIP spatial tokens [256, 1024] + multi-view DINO tokens; single view: CLIP || DINO
        c = {'crossattn': synth_input('prompt', (1, 256, 1024 if mv else 2048), args.seed), 'vector': synth_input('vec', (1, 768), args.seed)}
        if mv:
            c['concat'] = synth_input('mv', (1, args.num_mv_views, 256, 1024), args.seed)
        return c
    return {'crossattn': synth_input('prompt', (1, 77, args.context_dim), args.seed), 'vector': synth_input('vec', (1, 768), args.seed)}


def _save_ppm(path, frame):
    f0 = np.clip((frame.transpose(1, 2, 0) + 1) * 127.5, 0, 255).astype(np.uint8)
    with open(path, 'wb') as f:
        f.write(b'P6 %d %d 255\n' % (f0.shape[1], f0.shape[0]) + f0.tobytes())


def run(args):
    from . import parallel
    from .pipeline import T23DPipeline, FlowMatchingEngine, GuidedDiffusionEngine, render_video_given_triplane
    from .synth import orbit_cameras
    kind = validate(args)
    rank, local_rank, world = parallel.setup_dist()
    if not torch.cuda.is_available():
        raise SystemExit("this entry point runs the HIP path only (no CPU fallback)")
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    os.makedirs(args.logdir, exist_ok=True)
    if rank == 0:
        with open(os.path.join(args.logdir, 'args.json'), 'w') as f:
            json.dump(vars(args), f, indent=2)
    dit, ae, dec = build_models(args, dev, rank)
    parallel.broadcast_flat([p.data for p in dit.parameters()] + [p.data for p in dec.parameters()] + list(dec.buffers()), src=0)

    cond_all = load_conditioning(args, dev)
    P = cond_all['crossattn'].shape[0]
    cond_all = {k: v.repeat_interleave(args.num_samples, 0) for k, v in cond_all.items()}     # one condition x num_samples
    Bt = P * args.num_samples
    S = int(args.overwrite_diff_inp_size) if args.overwrite_diff_inp_size else args.diffusion_input_size
    torch.manual_seed(args.seed)                                                             # th.manual_seed, then randn(z_shape)
    z_all = torch.randn(Bt, (3 if args.roll_out else 1) * args.denoise_in_channels, S, S)
    lo, hi = parallel.shard_range(Bt, rank, world)
    z = z_all[lo:hi].to(dev)
    cond = {k: v[lo:hi].to(dev) for k, v in cond_all.items()}
    n_cam = 24 if kind == 'flow' else 40                                                      # camera[:24] / 40-view video
    cams = (torch.load(args.pose_path).float() if args.pose_path else orbit_cameras(max(args.num_views, 1)))
    cams = cams[:min(args.num_views, n_cam) if args.pose_path else args.num_views].to(dev)

    latent = torch.empty(0, *z_all.shape[1:], device=dev)
    if hi > lo:                                                   # a rank may own no sample (fewer samples than ranks)
        if kind == 'edm':
            eng = T23DPipeline(dit, ae, num_steps=args.sample_steps, cfg_scale=args.unconditional_guidance_scale,
                               triplane_scaling_divider=args.triplane_scaling_divider, img_size=args.image_size)
            latent = eng.sample_latent(z, cond, None)
        elif kind == 'flow':
            eng = FlowMatchingEngine(dit, ae, triplane_scaling_divider=args.triplane_scaling_divider, img_size=args.image_size,
                                     sampling_method=args.ode_method)
            latent = eng.sample(cond, None, batch_size=hi - lo, cfg_scale=args.unconditional_guidance_scale,
                                num_steps=args.sample_steps, zs=z)
        else:
            from .guided_diffusion import gaussian_diffusion as gd
            from .guided_diffusion.respace import SpacedDiffusion, space_timesteps
            spec = args.timestep_respacing or str(args.diffusion_steps)
            if args.use_ddim and not spec.startswith('ddim'):
                spec = 'ddim' + spec
            diff = SpacedDiffusion(use_timesteps=space_timesteps(args.diffusion_steps, spec),
                                   betas=gd.get_named_beta_schedule(args.noise_schedule, args.diffusion_steps))
            eng = GuidedDiffusionEngine(dit, ae, diff, triplane_scaling_divider=args.triplane_scaling_divider, img_size=args.image_size,
                                        diffusion_input_size=S)
            latent = eng.sample(cond, batch_size=hi - lo, use_ddim=args.use_ddim, noise=z, clip_denoised=args.clip_denoised,
                                unconditional_guidance_scale=args.unconditional_guidance_scale)
        out = render_video_given_triplane(latent.clone(), ae, cams, args.triplane_scaling_divider, export_mesh=args.export_mesh,
                                          mesh_size=args.mesh_grid, mesh_thres=args.mesh_thres, resolution=args.image_size,
                                          mesh_path=os.path.join(args.logdir, 'sample%d.obj').replace('%d', '{}') if args.export_mesh else None)
        if args.export_mesh:                      # mesh_path is formatted with the LOCAL index: rename to the global sample id
            for i in reversed(range(hi - lo)):
                src, dst = os.path.join(args.logdir, f'sample{i}.obj'), os.path.join(args.logdir, f'mesh_sample{lo + i}.obj')
                if os.path.exists(src):
                    os.replace(src, dst)
        frames = out['image_raw'].cpu().numpy()
        np.save(os.path.join(args.logdir, f'latent_rank{rank}.npy'), latent.cpu().numpy())
        np.save(os.path.join(args.logdir, f'frames_rank{rank}.npy'), frames)
        np.save(os.path.join(args.logdir, f'depth_rank{rank}.npy'), out['image_depth'].cpu().numpy())
        for i in range(hi - lo):
            _save_ppm(os.path.join(args.logdir, f'sample{lo + i}_view0.ppm'), frames[i, 0])
    lat_all = parallel.all_gather_cat(latent)                     # collective: EVERY rank calls it, also with zero rows
    if rank == 0:
        np.save(os.path.join(args.logdir, 'latents_all.npy'), lat_all.cpu().numpy())
        print(f"[rank0] {kind}: sampled {Bt} latents ({P} condition(s) x {args.num_samples}) on {world} GPU(s); outputs in {args.logdir}")
    parallel.barrier()
    return lat_all
