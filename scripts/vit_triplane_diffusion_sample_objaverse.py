#!/usr/bin/env python
"""Objaverse T23D / I23D sampling entry point - the flag surface and launch of the reference's
scripts/vit_triplane_diffusion_sample_objaverse.py (`torchrun --nproc_per_node=N scripts/vit_triplane_diffusion_sample_objaverse.py
--dit_model_arch DiT-L/2 --trainer_name sgm_legacy ...`, shell_scripts/final_release/inference/sample_obajverse_{t23d,i23d}_dit.sh)
on the HIP sampling path.  Body: ln3diff_amd/entry.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ln3diff_amd.entry import create_argparser, run  # noqa: E402

if __name__ == '__main__':
    args, unknown = create_argparser(objaverse=True).parse_known_args()
    if unknown:
        print(f"[entry] {len(unknown)} launcher flag(s) not used by the sampling path: {' '.join(u for u in unknown if u.startswith('--'))}")
    run(args)
