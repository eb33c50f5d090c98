#!/usr/bin/env python
"""ShapeNet / FFHQ sampling entry point - the flag surface of the reference's scripts/vit_triplane_diffusion_sample.py
(--trainer_name adm | vpsde_crossattn, --use_ddim, --timestep_respacing, --clip_denoised, --prompt, --cfg shapenet|ffhq,
--ddpm_model_path, --rec_model_path, --triplane_scaling_divider ...) driving the guided_diffusion engines
(SpacedDiffusion.p_sample_loop / ddim_sample_loop -> render_video_given_triplane) on the HIP path.  The reference runs these engines
over its U-Net denoiser, which is outside the hot path: the denoiser here is a registry DiT (--dit_model_arch).  Body: ln3diff_amd/entry.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ln3diff_amd.entry import create_argparser, run  # noqa: E402

if __name__ == '__main__':
    args, unknown = create_argparser(objaverse=False).parse_known_args()
    if unknown:
        print(f"[entry] {len(unknown)} launcher flag(s) not used by the sampling path: {' '.join(u for u in unknown if u.startswith('--'))}")
    run(args)
