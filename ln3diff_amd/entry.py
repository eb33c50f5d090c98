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


# Flags of the released launchers that describe the MODEL or the PREDICTION TYPE: accepted with the values those launchers pass, and
# refused (with the reason) when set to something this build does not implement - silently dropping them would change what is
# sampled.  (name: (default, {allowed values} or None = any, engines it matters for, message))
_RELEASED_DECODER = 'vit.vit_triplane.RodinSR_256_fusionv6_ConvQuant_liteSR_dinoInit3DAttn_SD_B_3L_C_withrollout_withSD_D_ditDecoder'
_CHECKED = {
    'mixed_prediction': (False, None, ('edm', 'flow', 'gd'), None),      # see validate(): the U-Net denoiser (--create_dit false) has it
    'predict_v': (False, None, ('gd',), None),            # see validate(): only the guided_diffusion engines read them
    'pred_type': ('eps', None, ('gd',), None),            # (guided_diffusion/script_util.py:36-37,84,682-686: predict_v -> ModelMeanType.V)
    'ae_classname': (_RELEASED_DECODER, {_RELEASED_DECODER}, ('edm', 'flow', 'gd'), "only the released decoder class is built"),
    'vae_p': (2, {2}, ('edm', 'flow', 'gd'), "the decoder tokeniser is built for vae_p = 2"),
    'denoise_out_channels': (4, None, ('edm', 'flow', 'gd'), None),
    'decoder_in_chans': (32, {32}, ('edm', 'flow', 'gd'), "tri-plane feature width 32 (OSGDecoder 32 -> 64 -> 4)"),
    'triplane_in_chans': (32, {32}, ('edm', 'flow', 'gd'), "tri-plane feature width 32"),
    'out_chans': (96, {96}, ('edm', 'flow', 'gd'), "3 planes x 32 channels"),
    'decoder_output_dim': (3, {3, 32}, ('edm', 'flow', 'gd'), "3 (Objaverse) or 32 (ShapeNet / FFHQ launchers; without the SR module only its first 3 colours are rendered)"),
    'patch_size': (14, None, (), None),                   # the VAE *encoder's* ViT patch size (DINO 14): encoder only, unused here
}


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
        ode_method='dopri5', use_ddim=False, clip_denoised=False, image_size=128, num_views=40 if objaverse else 24, export_mesh=False,
        mesh_grid=192, mesh_thres=10.0, logdir='./logs/sample', resume_checkpoint='', ddpm_model_path='', rec_model_path='',
        cond_path='', pose_path='', seed=41 if objaverse else 0, context_dim=768, learn_sigma=False, denoise_in_channels=4,
        diffusion_input_size=32, roll_out=objaverse, prompt=None, cfg='objverse_tuneray_aug_resolution_64_64_auto' if objaverse else 'shapenet',
        mv_input=False, num_mv_views=4, mv_dino_arch='vitl', clip_checkpoint='', dino_checkpoint='', tokenizer_dir='', image_path='',
        overwrite_diff_inp_size='', create_controlnet=False,
        # the U-Net denoiser of the ShapeNet / FFHQ launchers (guided_diffusion/script_util.py create_model).  r6: the second
        # script's defaults are the reference's model_and_diffusion_defaults (script_util.py:123,132: create_dit False, roll_out False) -
        # those launchers pass neither flag, so run verbatim they select the U-Net with a 12-channel latent, as in the reference
        create_dit=objaverse, num_channels=320, num_res_blocks=2, channel_mult='', attention_resolutions='4,2,1', num_heads=8,
        num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=True, use_spatial_transformer=True, transformer_depth=1,
        dropout=0.0, mixing_logit_init=-6.0)
    d.update(_IGNORED_DEFAULTS)
    d.update({k: v[0] for k, v in _CHECKED.items()})
    ap = argparse.ArgumentParser(allow_abbrev=False)
    add_dict_to_argparser(ap, d)
    ap.set_defaults(entry_objaverse=objaverse)             # which script's defaults these are (the default --prompt differs)
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
    unet = not args.create_dit
    if unet and kind != 'gd':
        raise SystemExit("--create_dit false (the U-Net denoiser) runs under the guided_diffusion engines: --trainer_name adm / ddpm / vpsde_crossattn")
    if args.mixed_prediction and not unet:
        raise SystemExit("--mixed_prediction true: only the U-Net denoiser (--create_dit false) defines a mixing_logit (the reference's DiT "
                         "classes have it commented out, dit/dit_models_xformers.py:767-772)")
    reg = I23D if args.i23d else T23D
    if not unet and args.dit_model_arch not in reg:
        other = T23D if args.i23d else I23D
        hint = " (that is an %s architecture: %s --i23d)" % (("T23D", "drop") if args.i23d else ("I23D", "pass")) if args.dit_model_arch in other else ""
        raise SystemExit(f"--dit_model_arch {args.dit_model_arch}: not in the {'I23D' if args.i23d else 'T23D'} registry {sorted(reg)}{hint}; "
                         "the U-Net denoiser is --create_dit false")
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
    for name, (default, allowed, kinds, why) in _CHECKED.items():
        val = getattr(args, name)
        if allowed is not None and kind in kinds and val not in allowed:
            raise SystemExit(f"--{name} {val}: {why} (supported: {sorted(allowed, key=str)})")
    if args.denoise_out_channels != args.denoise_in_channels:
        raise SystemExit(f"--denoise_out_channels {args.denoise_out_channels} != --denoise_in_channels {args.denoise_in_channels}: the "
                         "samplers update the latent in place with the network output (learn_sigma False)")
    if unet and (args.predict_v or args.pred_type == 'v') and not args.mixed_prediction:
        raise SystemExit("--predict_v true with the U-Net needs --mixed_prediction true: the reference's p_mean_variance only converts v to "
                         "eps inside its mixing branch (guided_diffusion/gaussian_diffusion.py:327-343, :399-400 asserts it)")
    if kind == 'gd' and not unet and (args.predict_v or args.pred_type not in ('eps', 'epsilon')):
        # create_gaussian_diffusion maps --predict_v to ModelMeanType.V (guided_diffusion/script_util.py:682-686); the guided_diffusion
        # engines of this build implement ModelMeanType.EPSILON.  (The released Objaverse launchers pass --predict_v True --pred_type v
        # but run the sgm / flow-matching engines, which never read them - accepted there.)
        raise SystemExit(f"--predict_v {args.predict_v} --pred_type {args.pred_type} with --trainer_name {args.trainer_name}: the "
                         "guided_diffusion engines of this build predict epsilon only (ModelMeanType.V / START_X are not built)")
    if 'PCD' in args.dit_model_arch:
        raise SystemExit(f"--dit_model_arch {args.dit_model_arch}: the point-cloud denoiser takes latents [B, N, C] and has no tri-plane "
                         "decode / render stage; this entry point draws z as [B, 3C, S, S] and renders tri-planes. Use the class "
                         "directly (ln3diff_amd.dit.dit_i23d.DiT_models) for point latents")
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
    if not args.create_dit:
        # guided_diffusion/script_util.py:255-451 create_model, U-Net branch (the ShapeNet / FFHQ launchers): the released VAE decoder of
        # those launchers (ViT-B decoder + 32-channel renderer + NearestConvSR) is not built - the latent is decoded by the released
        # Objaverse decoder class like every other denoiser's here
        from .guided_diffusion.unet import create_unet
        dit = create_unet(args.diffusion_input_size, args.num_channels, args.num_res_blocks, channel_mult=args.channel_mult,
                          learn_sigma=args.learn_sigma, attention_resolutions=args.attention_resolutions, num_heads=args.num_heads,
                          num_head_channels=args.num_head_channels, num_heads_upsample=args.num_heads_upsample,
                          use_scale_shift_norm=args.use_scale_shift_norm, dropout=args.dropout,
                          denoise_in_channels=args.denoise_in_channels, denoise_out_channels=args.denoise_out_channels,
                          mixed_prediction=args.mixed_prediction,
                          use_spatial_transformer=args.use_spatial_transformer, transformer_depth=args.transformer_depth,
                          context_dim=args.context_dim if args.use_spatial_transformer else None, mixing_logit_init=args.mixing_logit_init,
                          roll_out=args.roll_out)
        dit.embed_dim = {'DiT2-B/2': 768, 'DiT2-L/2': 1024, 'DiT2-XL/2': 1152}.get(args.arch_dit_decoder, 1024)     # the decoder tokeniser's width
    elif args.i23d:
        # multi-view denoisers cross-attend to the raw multi-view DINO tokens: their context width is the conditioner tower's
        # (--mv_dino_arch: vitl 1024, the released mv23d-plucker configs' vitb 768)
        ctx_dim = _MV_DINO_WIDTH[args.mv_dino_arch] if args.mv_input else 1024
        dit = I23D[args.dit_model_arch](context_dim=ctx_dim, pooling_ctx_dim=768, **common)
    else:
        dit = T23D[args.dit_model_arch](context_dim=args.context_dim, vit_blk=TextCondDiTBlock, **common)
    vit = DiT2_models[args.arch_dit_decoder](input_size=16, num_classes=0, learn_sigma=False, in_channels=dit.embed_dim,
                                             mixed_prediction=False, context_dim=None, roll_out=True, plane_n=3)
    dec = Dec(vit_decoder=vit, triplane_decoder=Triplane(img_resolution=args.image_size), cls_token=False, vae_p=2, ldm_z_channels=4,
              ldm_embed_dim=4)
    dit, dec = dit.to(dev), dec.to(dev)
    got = {'dit': None, 'decoder': None}                  # which file gave each component its weights
    if rank == 0:
        plan = [(args.resume_checkpoint, True, True), (args.ddpm_model_path, True, False), (args.rec_model_path, False, True)]
        for path, want_dit, want_dec in plan:
            if not path:
                continue
            # a joint file holds both components, the single-component files one: load what the file contains, strictly
            rep = load_checkpoint(path, dit=dit if want_dit else None, decoder=dec if want_dec else None, skip_absent=True)
            for comp in ('dit', 'decoder'):
                if comp in rep and rep[comp] != 'absent':
                    got[comp] = path
            if all(rep.get(c) in (None, 'absent') for c in ('dit', 'decoder')):
                raise SystemExit(f"{path}: holds neither denoiser nor decoder tensors under the known prefixes")
        for comp, mod, seed in (('dit', dit, 0), ('decoder', dec, 1)):
            if got[comp] is None:
                # no file for this component: synthetic weights, said out loud (constructor-initialised weights - adaLN-zero etc. -
                # would produce all-zero denoiser outputs / empty renders without an error)
                print(f"[entry] WARNING: no checkpoint provides the {comp}: filling it with SYNTHETIC random weights (seed {seed})")
                fill_module_random_(mod, seed, dev)
                if comp == 'decoder':
                    dec.triplane_decoder.decoder.net[2].bias.data[0] += 4.0
    return dit, AE(None, dec, args.image_size), dec, got


_MV_DINO_WIDTH = {'vits': 384, 'vitb': 768, 'vitl': 1024}


def load_conditioning(args, dev, objaverse=True):
    """({'crossattn', 'vector'[, 'concat']} for P prompts, source).  Sources, in order: --cond_path tensors; the HIP conditioners on
    --prompt (CLIP-L text tower, T23D) / --image_path (OpenCLIP ViT-L/14 + DINOv2 ViT-L/14-reg, I23D) when their checkpoints are
    given; synthetic tensors of the right shapes ONLY when no prompt / image was asked for.  A prompt or an image that cannot be
    encoded is refused: sampling from noise conditioning while reporting a normal run would be silently wrong."""
    from .synth import synth_input
    from .checkpoint import load_checkpoint
    if args.cond_path:
        raw = torch.load(args.cond_path) if args.cond_path.endswith('.pt') else dict(np.load(args.cond_path))
        return {k: torch.as_tensor(v).float() for k, v in raw.items()}, f'tensors from {args.cond_path}'
    # --prompt defaults to None: "not given" (the reference scripts' defaults - '' / 'a red chair' - then only label the synthetic
    # run); ANY explicit --prompt, also one equal to a script default, must be encoded or is refused
    if args.image_path:
        if not args.i23d:
            raise SystemExit("--image_path conditions the I23D models: pass --i23d true (and --trainer_name flow_matching)")
        if args.mv_input:
            # multi-view denoisers (DiT-PixArt-MV-*): --image_path is an .npz with 'img' [T, 3, H, W] (or [T, H, W, 3] uint8) and 'c'
            # [T, 25] (camera-to-world 4x4 + normalised intrinsics 3x3 per view); the first --num_mv_views views condition the sample
            # through FrozenDinov2ImageEmbedderMVPlucker (released mv23d-plucker configs), the first view through OpenCLIP for the
            # non-noClip denoisers
            if not args.image_path.endswith('.npz'):
                raise SystemExit("--image_path with a multi-view (MV) denoiser expects an .npz holding 'img' [T, 3, H, W] and 'c' [T, 25]")
            from .dit.dit_i23d import MV_NOCLIP_ARCHS
            noclip = args.dit_model_arch in MV_NOCLIP_ARCHS
            if not args.dino_checkpoint or (not noclip and not args.clip_checkpoint):
                raise SystemExit("--image_path with an MV denoiser needs --dino_checkpoint (embedder_FrozenDinov2ImageEmbedderMVPlucker*.pt)"
                                 + ("" if noclip else " and --clip_checkpoint (open_clip ViT-L/14 visual tower)") +
                                 ": without them the views cannot be encoded (refusing to sample from synthetic conditioning)")
            from .sgm.image_encoders import FrozenOpenCLIPImageEmbedder, FrozenDinov2ImageEmbedderMVPlucker, MV23DConditioner
            raw = np.load(args.image_path)
            img = torch.as_tensor(raw['img'])
            if img.dtype == torch.uint8:
                img = img.float() / 127.5 - 1.0
            if img.ndim == 4 and img.shape[-1] == 3:
                img = img.permute(0, 3, 1, 2)
            cam = torch.as_tensor(raw['c']).float().reshape(img.shape[0], 25)
            if img.shape[0] < args.num_mv_views:
                raise SystemExit(f"--image_path holds {img.shape[0]} views, the denoiser is conditioned on --num_mv_views {args.num_mv_views}")
            dino = FrozenDinov2ImageEmbedderMVPlucker(arch=args.mv_dino_arch, n_cond_frames=args.num_mv_views, device=str(dev))
            load_checkpoint(args.dino_checkpoint, conditioner=dino)
            clip = None
            if not noclip:
                clip = FrozenOpenCLIPImageEmbedder(device=str(dev), output_tokens=True)
                load_checkpoint(args.clip_checkpoint, conditioner=clip.model)
                clip = clip.to(dev)
            c = MV23DConditioner(dino.to(dev), clip)({'img': img.float()[None].to(dev), 'c': cam[None].to(dev)})
            return {k: v.float().cpu() for k, v in c.items()}, f'MV23DConditioner({args.image_path}, {args.num_mv_views} views)'
        if not (args.clip_checkpoint and args.dino_checkpoint):
            raise SystemExit("--image_path needs --clip_checkpoint (open_clip ViT-L/14 visual tower) and --dino_checkpoint (DINOv2 "
                             "ViT-L/14-reg): without them the image cannot be encoded (refusing to sample from synthetic conditioning)")
        from .sgm.image_encoders import FrozenOpenCLIPImageEmbedder, FrozenDinov2ImageEmbedder, I23DConditioner
        clip, dino = FrozenOpenCLIPImageEmbedder(device=str(dev), output_tokens=True), FrozenDinov2ImageEmbedder(device=str(dev))
        load_checkpoint(args.clip_checkpoint, conditioner=clip.model)        # strict: an unmapped / missing key raises
        load_checkpoint(args.dino_checkpoint, conditioner=dino.model)
        img = _read_image(args.image_path).to(dev)
        c = I23DConditioner(clip.to(dev), dino.to(dev))(img)
        return {k: v.float().cpu() for k, v in c.items()}, f'I23DConditioner({args.image_path})'
    if args.prompt is not None:
        if args.i23d:
            raise SystemExit("--prompt conditions the T23D models; the I23D models take --image_path / --cond_path")
        if not args.clip_checkpoint:
            raise SystemExit(f"--prompt {args.prompt!r} needs --clip_checkpoint (CLIP-L text tower; plus --tokenizer_dir with vocab.json + "
                             "merges.txt): without it the prompt cannot be encoded (refusing to sample from synthetic conditioning)")
        from .sgm.encoders import FrozenCLIPEmbedder
        enc = FrozenCLIPEmbedder(device=str(dev), always_return_pooled=True, tokenizer_dir=args.tokenizer_dir or None)
        load_checkpoint(args.clip_checkpoint, conditioner=enc)
        z, pooled = enc.to(dev)([args.prompt])
        return {'crossattn': z.float().cpu(), 'vector': pooled.float().cpu()}, f'FrozenCLIPEmbedder({args.prompt!r})'
    if args.i23d:
        mv = args.mv_input                # MVCond: CLIP spatial tokens [256, 1024] + multi-view DINO tokens; single view: CLIP || DINO
        # the plain DiT_I23D ('DiT-L/2' ... of the I23D registry) embeds the pooled token with clip_text_proj(context_dim = 1024)
        vdim = 768 if 'PixArt' in args.dit_model_arch else 1024
        c = {'crossattn': synth_input('prompt', (1, 256, 1024 if mv else 2048), args.seed), 'vector': synth_input('vec', (1, vdim), args.seed)}
        if mv:
            c['concat'] = synth_input('mv', (1, args.num_mv_views, 256, _MV_DINO_WIDTH[args.mv_dino_arch]), args.seed)
        return c, 'synthetic'
    return ({'crossattn': synth_input('prompt', (1, 77, args.context_dim), args.seed), 'vector': synth_input('vec', (1, 768), args.seed)},
            'synthetic')


def _read_image(path):
    """[1, 3, H, W] in [-1, 1] from a .npy / .pt array ([H, W, 3] uint8 or [3, H, W] float), a binary PPM (P6) or any image Pillow
    decodes (alpha composited on white); the preprocessing
    to 224^2 and the CLIP / DINO normalisation happen in the embedders (sgm/image_encoders.py)."""
    if path.endswith('.npy') or path.endswith('.pt'):
        a = torch.as_tensor(np.load(path) if path.endswith('.npy') else torch.load(path))
    elif path.endswith('.ppm'):
        with open(path, 'rb') as f:
            toks = []
            while len(toks) < 4:
                line = f.readline()
                if not line.startswith(b'#'):
                    toks += line.split()
            assert toks[0] == b'P6' and int(toks[3]) == 255, "binary 8-bit PPM expected"
            w, h = int(toks[1]), int(toks[2])
            a = torch.frombuffer(bytearray(f.read(w * h * 3)), dtype=torch.uint8).reshape(h, w, 3)
    else:
        # encoded images (png / jpg / webp ...) through Pillow when the environment has it, as the reference's loaders do
        # (the released demo reads RGBA pngs and composites them on white: datasets/g_buffer_objaverse.py preprocessing)
        try:
            from PIL import Image
        except ImportError:
            raise SystemExit(f"--image_path {path}: .npy / .pt arrays and binary .ppm are read without a codec; encoded images need Pillow")
        im = Image.open(path)
        if im.mode in ('RGBA', 'LA', 'P'):
            im = im.convert('RGBA')
            bg = Image.new('RGBA', im.size, (255, 255, 255, 255))
            im = Image.alpha_composite(bg, im)
        a = torch.from_numpy(np.asarray(im.convert('RGB')).copy())
    if a.dtype == torch.uint8:
        a = a.float() / 127.5 - 1.0
    a = a.float()
    if a.ndim == 3 and a.shape[-1] == 3:
        a = a.permute(2, 0, 1)
    if a.ndim == 3:
        a = a[None]
    if a.ndim != 4 or a.shape[1] != 3:
        raise SystemExit(f"--image_path {path}: expected an RGB image, got shape {tuple(a.shape)}")
    return a


def _save_ppm(path, frame):
    f0 = np.clip((frame.transpose(1, 2, 0) + 1) * 127.5, 0, 255).astype(np.uint8)
    with open(path, 'wb') as f:
        f.write(b'P6 %d %d 255\n' % (f0.shape[1], f0.shape[0]) + f0.tobytes())


def run(args, objaverse=None):
    objaverse = getattr(args, 'entry_objaverse', True) if objaverse is None else objaverse
    from . import parallel
    from .pipeline import T23DPipeline, FlowMatchingEngine, GuidedDiffusionEngine, render_pairs
    from .synth import orbit_cameras
    kind = validate(args)
    rank, local_rank, world = parallel.setup_dist()
    if not torch.cuda.is_available():
        raise SystemExit("this entry point runs the HIP path only (no CPU fallback)")
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    os.makedirs(args.logdir, exist_ok=True)
    dit, ae, dec, weights_from = build_models(args, dev, rank)
    parallel.broadcast_flat([p.data for p in dit.parameters()] + [p.data for p in dec.parameters()] + list(dec.buffers()), src=0)

    # conditioning is produced ONCE, on rank 0 (checkpoint loads, image / prompt encoding), and broadcast; a failure there is
    # broadcast too, so that every rank leaves with the same message instead of waiting in the next collective
    res = [None]
    if rank == 0:
        try:
            res = [load_conditioning(args, dev, objaverse) + (None,)]
        except BaseException as e:
            res = [(None, None, f"{type(e).__name__}: {e}")]
    res = parallel.broadcast_object(res[0], src=0)
    if res[2] is not None:
        raise SystemExit("conditioning failed on rank 0: " + res[2])
    cond_all, cond_src = res[0], res[1]
    if rank == 0:
        meta = dict(vars(args), conditioning=cond_src, weights={k: (v or 'synthetic') for k, v in weights_from.items()})
        with open(os.path.join(args.logdir, 'args.json'), 'w') as f:
            json.dump(meta, f, indent=2)
    P = next(iter(cond_all.values())).shape[0]                                             # the noClip multi-view denoisers take 'concat' only
    cond_all = {k: v.repeat_interleave(args.num_samples, 0) for k, v in cond_all.items()}     # one condition x num_samples
    Bt = P * args.num_samples
    S = int(args.overwrite_diff_inp_size) if args.overwrite_diff_inp_size else args.diffusion_input_size
    torch.manual_seed(args.seed)                                                             # th.manual_seed, then randn(z_shape)
    z_all = torch.randn(Bt, (3 if args.roll_out else 1) * args.denoise_in_channels, S, S)
    n_cam = 24 if kind == 'flow' else 40                                                      # camera[:24] / 40-view video
    cams = (torch.load(args.pose_path).float() if args.pose_path else orbit_cameras(max(args.num_views, 1)))
    cams = cams[:min(args.num_views, n_cam) if args.pose_path else args.num_views].to(dev)
    V = cams.shape[0]

    if kind == 'edm':
        eng = T23DPipeline(dit, ae, num_steps=args.sample_steps, cfg_scale=args.unconditional_guidance_scale,
                           triplane_scaling_divider=args.triplane_scaling_divider, img_size=args.image_size)
    elif kind == 'flow':
        # --ode_method defaults to dopri5 like the reference's sample_ode (torchdiffeq dopri5, atol 1e-6, rtol 1e-3;
        # transport/transport.py:377); the benchmark configurations ("50 steps") pass --ode_method euler
        eng = FlowMatchingEngine(dit, ae, triplane_scaling_divider=args.triplane_scaling_divider, img_size=args.image_size,
                                 sampling_method=args.ode_method)
    else:
        from .guided_diffusion import gaussian_diffusion as gd
        from .guided_diffusion.respace import SpacedDiffusion, space_timesteps
        spec = args.timestep_respacing or str(args.diffusion_steps)
        if args.use_ddim and not spec.startswith('ddim'):
            spec = 'ddim' + spec
        v_pred = (not args.create_dit) and (args.predict_v or args.pred_type == 'v')          # create_gaussian_diffusion: predict_v -> ModelMeanType.V
        diff = SpacedDiffusion(use_timesteps=space_timesteps(args.diffusion_steps, spec),
                               betas=gd.get_named_beta_schedule(args.noise_schedule, args.diffusion_steps),
                               model_mean_type=gd.ModelMeanType.V if v_pred else gd.ModelMeanType.EPSILON)
        eng = GuidedDiffusionEngine(dit, ae, diff, triplane_scaling_divider=args.triplane_scaling_divider, img_size=args.image_size,
                                    diffusion_input_size=S)

    def sample_fn(lo, hi):                                        # a rank may own no sample (fewer samples than ranks)
        if hi <= lo:
            return torch.empty(0, *z_all.shape[1:], device=dev)
        z = z_all[lo:hi].to(dev)
        cond = {k: v[lo:hi].to(dev) for k, v in cond_all.items()}
        if kind == 'edm':
            return eng.sample_latent(z, cond, None)
        if kind == 'flow':
            return eng.sample(cond, None, batch_size=hi - lo, cfg_scale=args.unconditional_guidance_scale, num_steps=args.sample_steps, zs=z)
        return eng.sample(cond, batch_size=hi - lo, use_ddim=args.use_ddim, noise=z, clip_denoised=args.clip_denoised,
                          unconditional_guidance_scale=args.unconditional_guidance_scale)

    def render_fn(latent_all, pairs):                             # this rank's (sample, view) pairs: views are shared out when Bt < world
        return render_pairs(latent_all, ae, cams, pairs, args.triplane_scaling_divider, resolution=args.image_size, noise_seed=args.seed)

    lat_all, frames, pairs = parallel.sharded_step(sample_fn, render_fn, Bt, V, rank, world)
    lo, hi = parallel.shard_range(Bt, rank, world)
    mesh_err = None
    if args.export_mesh and hi > lo:
        # meshes go with the SAMPLE shard; the file name carries the global sample id from the start (r2 wrote `sample{local}.obj`
        # on every rank and renamed afterwards: two ranks raced on the same names)
        from .mesh import mesh_from_grid
        mine = lat_all[lo:hi].clone()
        mine *= args.triplane_scaling_divider
        d = {'latent_normalized_2Ddiffusion': mine}
        d.update(ae(latent=d, behaviour='decode_after_vae_no_render'))
        grid = ae(latent=d, grid_size=args.mesh_grid, behaviour='triplane_decode_grid')
        for i in range(hi - lo):
            path = os.path.join(args.logdir, f'mesh_sample{lo + i}.obj')
            mesh_from_grid(ae.decoder, d, grid['sigma'][i], args.mesh_grid, args.mesh_thres, sample_index=i, path=path)
            if not os.path.exists(path):
                mesh_err = f"mesh export of sample {lo + i} produced no file at {path}"
                break
    parallel.agree_ok(mesh_err)                               # collective: every rank raises when any rank failed
    # frames_rank{r}.npy / depth_rank{r}.npy: ALWAYS the flat pair list [P, 3 | 1, R, R]; pairs_rank{r}.npy = the (sample, view) of
    # every frame (one layout whatever the sharding: r3 switched between [S, V, ...] and [P, ...])
    np.save(os.path.join(args.logdir, f'frames_rank{rank}.npy'), frames['image_raw'].cpu().numpy())
    np.save(os.path.join(args.logdir, f'depth_rank{rank}.npy'), frames['image_depth'].cpu().numpy())
    np.save(os.path.join(args.logdir, f'pairs_rank{rank}.npy'), frames['pair_index'].cpu().numpy())      # [P, 2] = (sample, view) of every frame
    np.save(os.path.join(args.logdir, f'latent_rank{rank}.npy'), lat_all[lo:hi].cpu().numpy())
    fr = frames['image_raw'].cpu().numpy()
    for j, (smp, view) in enumerate(frames['pair_index'].tolist()):
        if view == 0:
            _save_ppm(os.path.join(args.logdir, f'sample{smp}_view0.ppm'), fr[j])
    if rank == 0:
        np.save(os.path.join(args.logdir, 'latents_all.npy'), lat_all.cpu().numpy())
        print(f"[rank0] {kind}: sampled {Bt} latents ({P} condition(s) x {args.num_samples}) and rendered {Bt * V} views on {world} GPU(s); "
              f"conditioning: {cond_src}; weights: " + ", ".join(f"{k}={'checkpoint' if v else 'SYNTHETIC'}" for k, v in weights_from.items())
              + f"; outputs in {args.logdir}")
    parallel.barrier()
    return lat_all
