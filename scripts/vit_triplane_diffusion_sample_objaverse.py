#!/usr/bin/env python
"""Entry point with the flag surface of the reference's scripts/vit_triplane_diffusion_sample_objaverse.py
(launched the same way: `torchrun --nproc_per_node=N scripts/vit_triplane_diffusion_sample_objaverse.py --flags`,
reference shell_scripts/final_release/inference/sample_obajverse_{t23d,i23d}_dit.sh), running the HIP sampling path.

Differences that matter to a user:
  * all ranks sample (the batch of --num_samples x prompts is sharded over ranks); the reference samples on rank 0 only
    (scripts/vit_triplane_diffusion_sample_objaverse.py:170);
  * the CLIP / DINO conditioners are outside the hot path: conditioning comes from --cond_path
    (a .pt/.npz with 'crossattn' [P,77,768] (+ 'vector') for T23D, 'crossattn' [P,256,2048] + 'vector' [P,768] for I23D)
    or is synthesised (--synthetic_cond, default when no path is given);
  * weights: --resume_checkpoint (a .safetensors/.pt state-dict with `ddpm_model.*` / `rec_model.decoder.*` prefixes or
    bare keys) or deterministic synthetic weights;
  * outputs under --logdir: latents .npy, per-view frames .npy (+ .ppm), sigma/rgb grid .npy when --export_mesh.
Training-only flags of the reference launcher are accepted and ignored.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def str2bool(v):
    if isinstance(v, bool):
        return v
    return v.lower() in ("yes", "true", "t", "y", "1")


def create_argparser():
    ap = argparse.ArgumentParser(allow_abbrev=False)
    ap.add_argument("--dit_model_arch", default="DiT-L/2")
    ap.add_argument("--arch_dit_decoder", default="DiT2-L/2")
    ap.add_argument("--i23d", type=str2bool, default=False)
    ap.add_argument("--trainer_name", default="sgm_legacy")          # sgm_legacy (EulerEDM) | flow_matching | ddpm
    ap.add_argument("--num_samples", type=int, default=4)
    ap.add_argument("--unconditional_guidance_scale", type=float, default=6.5)
    ap.add_argument("--triplane_scaling_divider", type=float, default=0.96806)
    ap.add_argument("--timestep_respacing", default="ddim250")
    ap.add_argument("--sample_steps", type=int, default=250)
    ap.add_argument("--ode_method", default="euler")
    ap.add_argument("--image_size", type=int, default=128, help="render resolution")
    ap.add_argument("--num_views", type=int, default=40)
    ap.add_argument("--export_mesh", type=str2bool, default=False)
    ap.add_argument("--mesh_grid", type=int, default=192)
    ap.add_argument("--logdir", default="./logs/sample")
    ap.add_argument("--resume_checkpoint", default="")
    ap.add_argument("--cond_path", default="")
    ap.add_argument("--pose_path", default="", help="[V,25] camera file (e.g. the reference's assets/objv_eval_pose.pt)")
    ap.add_argument("--seed", type=int, default=41)
    ap.add_argument("--context_dim", type=int, default=768)
    ap.add_argument("--learn_sigma", type=str2bool, default=False)
    ap.add_argument("--denoise_in_channels", type=int, default=4)
    ap.add_argument("--diffusion_input_size", type=int, default=32)
    ap.add_argument("--roll_out", type=str2bool, default=True)
    ap.add_argument("--prompt", default="")
    return ap


def load_checkpoint(path, dit, dec):
    from ln3diff_amd.checkpoint import load_checkpoint as _load
    return _load(path, dit=dit, decoder=dec)


def main():
    args, ignored = create_argparser().parse_known_args()
    from ln3diff_amd import parallel
    from ln3diff_amd.pipeline import T23DPipeline, TRIPLANE_SCALING_DIVIDER
    from ln3diff_amd.synth import orbit_cameras, synth_input, synth_state_dict
    from bench import build_models
    rank, local_rank, world = parallel.setup_dist()
    if not torch.cuda.is_available():
        raise SystemExit("this entry point runs the HIP path only (no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    os.makedirs(args.logdir, exist_ok=True)

    if args.i23d:
        from ln3diff_amd.dit.dit_i23d import DiT_models as I23D
        from ln3diff_amd.synth import fill_module_random_
        _, dec = build_models(dev, "DiT-B/2", args.arch_dit_decoder, fill=(rank == 0 and not args.resume_checkpoint))
        dit = I23D[args.dit_model_arch](input_size=args.diffusion_input_size, num_classes=0, learn_sigma=args.learn_sigma,
                                        in_channels=args.denoise_in_channels, context_dim=1024, roll_out=True,
                                        pooling_ctx_dim=768).to(dev)
        if rank == 0 and not args.resume_checkpoint:
            fill_module_random_(dit, 0, dev)
    else:
        dit, dec = build_models(dev, args.dit_model_arch, args.arch_dit_decoder,
                                fill=(rank == 0 and not args.resume_checkpoint))
    if args.resume_checkpoint and rank == 0:
        load_checkpoint(args.resume_checkpoint, dit, dec)
    parallel.broadcast_flat([p.data for p in dit.parameters()] + [p.data for p in dec.parameters()], src=0)

    # conditioning
    if args.cond_path:
        raw = torch.load(args.cond_path) if args.cond_path.endswith(".pt") else dict(np.load(args.cond_path))
        cond_all = {k: torch.as_tensor(v).float() for k, v in raw.items()}
    else:
        shp = (1, 256, 2048) if args.i23d else (1, 77, args.context_dim)
        cond_all = {"crossattn": synth_input("prompt", shp, args.seed), "vector": synth_input("vec", (1, 768), args.seed)}
    P = cond_all["crossattn"].shape[0]
    cond_all = {k: v.repeat_interleave(args.num_samples, 0) for k, v in cond_all.items()}   # eval_cldm :476
    Bt = P * args.num_samples
    torch.manual_seed(args.seed)                                    # th.manual_seed(41), sgm_DiffusionEngine.py:457
    z_all = torch.randn(Bt, 3 * args.denoise_in_channels, args.diffusion_input_size, args.diffusion_input_size)
    lo, hi = parallel.shard_range(Bt, rank, world)
    z = z_all[lo:hi].to(dev)
    cond = {k: v[lo:hi].to(dev) for k, v in cond_all.items()}
    uc = {k: torch.zeros_like(v) for k, v in cond.items()}
    cams = (torch.load(args.pose_path).float() if args.pose_path else orbit_cameras(args.num_views))[:args.num_views].to(dev)

    pipe = T23DPipeline(dit, dec, num_steps=args.sample_steps, cfg_scale=args.unconditional_guidance_scale)
    if hi > lo:
        if args.i23d or args.trainer_name == "flow_matching":
            from ln3diff_amd.transport import Sampler, create_transport
            ctx = {k: torch.cat([cond[k], uc[k]], 0) for k in cond}                 # flow matching: [c, uc]
            cache = dit.prepare_context(ctx)
            fn = Sampler(create_transport(snr_type="lognorm")).sample_ode(sampling_method=args.ode_method,
                                                                          num_steps=args.sample_steps)
            latent = fn(torch.cat([z, z]), dit.forward_with_cfg, return_trajectory=False, context_cache=cache,
                        cfg_scale=args.unconditional_guidance_scale)[-1].chunk(2)[0].contiguous()
        elif args.trainer_name == "ddpm":
            from ln3diff_amd.guided_diffusion import gaussian_diffusion as gd
            from ln3diff_amd.guided_diffusion.respace import SpacedDiffusion, space_timesteps
            diff = SpacedDiffusion(use_timesteps=space_timesteps(1000, args.timestep_respacing.replace("ddim", "")),
                                   betas=gd.get_named_beta_schedule("linear", 1000))
            latent = diff.p_sample_loop(dit, tuple(z.shape), cond=cond["crossattn"], noise=z, clip_denoised=False)
        else:
            latent = pipe.sample_latent(z, cond, uc)
        dec_out = pipe.decode(latent, want_nchw=False)
        img = pipe.render(dec_out, cams, args.image_size)
        np.save(os.path.join(args.logdir, f"latent_rank{rank}.npy"), latent.cpu().numpy())
        frames = img["image_raw"].cpu().numpy()
        np.save(os.path.join(args.logdir, f"frames_rank{rank}.npy"), frames)
        f0 = np.clip((frames[0, 0].transpose(1, 2, 0) + 1) * 127.5, 0, 255).astype(np.uint8)
        with open(os.path.join(args.logdir, f"sample{lo}_view0.ppm"), "wb") as f:
            f.write(b"P6 %d %d 255\n" % (f0.shape[1], f0.shape[0]) + f0.tobytes())
        if args.export_mesh:
            from ln3diff_amd.mesh import export_mesh           # sigma grid -> iso-surface (thr 10) -> coloured .obj
            for i in range(latent.shape[0]):
                export_mesh(dec, dec_out, os.path.join(args.logdir, f"sample{lo + i}.obj"), grid_size=args.mesh_grid,
                            thr=10.0, sample_index=i)
        lat_all = parallel.all_gather_cat(latent)
        if rank == 0:
            np.save(os.path.join(args.logdir, "latents_all.npy"), lat_all.cpu().numpy())
            print(f"[rank0] sampled {Bt} latents on {world} GPU(s); outputs in {args.logdir}; ignored flags: {len(ignored)}")
    parallel.barrier()


if __name__ == "__main__":
    main()
