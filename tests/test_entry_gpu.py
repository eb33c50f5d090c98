"""The two entry points end to end on the GPU (small registry models, few steps): every engine of the trainer table samples, renders,
exports a mesh and writes its outputs; results do not depend on how the batch is sharded."""
import os

import numpy as np
import pytest
import torch

from ln3diff_amd.entry import create_argparser, run

pytestmark = pytest.mark.gpu

SMALL = "--arch_dit_decoder DiT2-B/2 --num_samples 2 --sample_steps 4 --image_size 32 --num_views 2 --mesh_grid 24"


@pytest.mark.parametrize("objaverse,flags", [
    (True, "--dit_model_arch DiT-B/2 --trainer_name sgm_legacy --export_mesh true --mesh_thres 4.0"),
    (True, "--dit_model_arch DiT-PixArt-B/2 --i23d true --trainer_name flow_matching --unconditional_guidance_scale 4.0"),
    (True, "--dit_model_arch DiT-PixArt-MV-B/2 --i23d true --trainer_name flow_matching --num_mv_views 2"),
    (True, "--dit_model_arch DiT-B/2 --i23d true --trainer_name flow_matching --unconditional_guidance_scale 4.0"),      # plain DiT_I23D
    (True, "--dit_model_arch DiT-PixelArt-B/2 --trainer_name flow_matching --unconditional_guidance_scale 4.0"),   # T23D flow matching
    (False, "--create_dit true --roll_out true --dit_model_arch DiT-B/2 --trainer_name adm --timestep_respacing 4"),
    (False, "--create_dit true --roll_out true --dit_model_arch DiT-B/2 --trainer_name vpsde_crossattn --use_ddim true --timestep_respacing ddim4 --unconditional_guidance_scale 3.0"),
    # the U-Net denoiser (ShapeNet launcher flags at a small width): v-prediction + mixed prediction, DDIM with CFG / ancestral sampling
    (False, "--create_dit false --trainer_name vpsde_crossattn --num_channels 128 --num_res_blocks 1 --num_heads 4 --channel_mult 1,2 "
            "--attention_resolutions 32,16 --denoise_in_channels 12 --denoise_out_channels 12 --roll_out false --predict_v true --pred_type v "
            "--mixed_prediction true --use_ddim true --timestep_respacing ddim4 --unconditional_guidance_scale 2.0"),
    # the second script's own defaults select the U-Net (create_dit False, roll_out False: guided_diffusion/script_util.py:123,132)
    (False, "--trainer_name adm --num_channels 128 --num_res_blocks 1 --num_heads 4 --channel_mult 1,2 "
            "--attention_resolutions 16 --use_spatial_transformer false --timestep_respacing 4 --denoise_in_channels 12 --denoise_out_channels 12"),
])
def test_entry_points_run(hip_lib, tmp_path, objaverse, flags):
    args = create_argparser(objaverse).parse_args((SMALL + " " + flags + f" --logdir {tmp_path}").split())
    lat = run(args)
    assert lat.shape == (2, 12, 32, 32) and torch.isfinite(lat).all()
    frames = np.load(tmp_path / "frames_rank0.npy")
    assert frames.shape == (4, 3, 32, 32) and np.isfinite(frames).all()            # flat (sample, view) pair list, pairs_rank0.npy names them
    assert np.load(tmp_path / "pairs_rank0.npy").tolist() == [[0, 0], [0, 1], [1, 0], [1, 1]]
    assert np.array_equal(np.load(tmp_path / "latents_all.npy"), lat.cpu().numpy())
    assert os.path.exists(tmp_path / "sample1_view0.ppm") and os.path.exists(tmp_path / "args.json")
    if args.export_mesh:
        assert os.path.exists(tmp_path / "mesh_sample0.obj") and os.path.exists(tmp_path / "mesh_sample1.obj")   # content: test_mesh_gpu.py
    # the divider is applied: a different --triplane_scaling_divider changes the frames, not the latents
    args2 = create_argparser(objaverse).parse_args((SMALL + " " + flags + f" --logdir {tmp_path}/b --triplane_scaling_divider 0.5").split())
    args2.export_mesh = False
    lat2 = run(args2)
    assert torch.equal(lat2, lat)
    assert not np.array_equal(np.load(tmp_path / "b" / "frames_rank0.npy"), frames)


def test_config3_xl2_text_cond_end_to_end(hip_lib, tmp_path):
    """BASELINE configs[3] on one GPU's share at reduced steps: DiT-XL/2 text-conditioned EulerEDM + CFG -> decode -> views."""
    args = create_argparser(True).parse_args(("--dit_model_arch DiT-XL/2 --trainer_name sgm_legacy --num_samples 2 --sample_steps 3 "
                                              f"--image_size 64 --num_views 3 --logdir {tmp_path}").split())
    lat = run(args)
    assert lat.shape == (2, 12, 32, 32) and torch.isfinite(lat).all()
    assert np.load(tmp_path / "frames_rank0.npy").shape == (6, 3, 64, 64)


def test_config4_i23d_512_24cams_mesh_end_to_end(hip_lib, tmp_path):
    """BASELINE configs[4] on one GPU's share at reduced steps: DiT-PixArt-L/2 I23D flow matching, 24 cameras @ 512^2, marching-cubes
    mesh export."""
    args = create_argparser(True).parse_args(("--dit_model_arch DiT-PixArt-L/2 --i23d true --trainer_name flow_matching --num_samples 1 "
                                              "--sample_steps 4 --ode_method euler --unconditional_guidance_scale 4.0 --image_size 512 --num_views 24 "
                                              f"--export_mesh true --mesh_grid 96 --mesh_thres 4.0 --logdir {tmp_path}").split())
    lat = run(args)
    assert lat.shape == (1, 12, 32, 32) and torch.isfinite(lat).all()
    frames = np.load(tmp_path / "frames_rank0.npy")
    depth = np.load(tmp_path / "depth_rank0.npy")
    print('config4 frames', frames.shape, 'finite', bool(np.isfinite(frames).all()), 'absmax', float(np.nanmax(np.abs(frames))), 'latent std', float(lat.std()))
    assert frames.shape == (24, 3, 512, 512) and np.isfinite(frames).all() and np.abs(frames).max() <= 1.0 + 2e-3
    assert depth.shape == (24, 1, 512, 512) and np.isfinite(depth).all()
    assert os.path.exists(tmp_path / "mesh_sample0.obj")


def test_multiview_image_path_runs_the_plucker_conditioner(hip_lib, tmp_path):
    """r4 (f)3: --image_path with a multi-view denoiser encodes posed views through FrozenDinov2ImageEmbedderMVPlucker (synthetic
    embedder checkpoint in the reference's own key layout) instead of asking for --cond_path tensors; the run records its source."""
    import json
    from ln3diff_amd.sgm.image_encoders import FrozenDinov2ImageEmbedderMVPlucker
    from ln3diff_amd.synth import synth_vit_state_dict, synth_input
    emb = FrozenDinov2ImageEmbedderMVPlucker(arch='vitb', n_cond_frames=2, device='cpu')
    sd = {'model.' + k: v for k, v in synth_vit_state_dict({k[6:]: tuple(v.shape) for k, v in emb.state_dict().items()}, 3).items()}
    torch.save(sd, tmp_path / "embedder_FrozenDinov2ImageEmbedderMVPlucker0000001.pt")
    cams = np.zeros((3, 25), np.float32)
    for t in range(3):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = (0.3 * t, 0.1, -1.8)
        cams[t, :16], cams[t, 16:] = m.reshape(-1), (1.1, 0, 0.5, 0, 1.1, 0.5, 0, 0, 1)
    np.savez(tmp_path / "views.npz", img=synth_input('mvimg', (3, 3, 224, 224), 5).clamp(-1, 1).numpy(), c=cams)
    flags = ("--dit_model_arch DiT-PixArt-MV-L/2 --i23d true --trainer_name flow_matching --num_mv_views 2 --mv_dino_arch vitb --num_samples 1 "
             "--sample_steps 3 --ode_method euler --unconditional_guidance_scale 4.0 --image_size 32 --num_views 2 "
             f"--image_path {tmp_path}/views.npz --dino_checkpoint {tmp_path}/embedder_FrozenDinov2ImageEmbedderMVPlucker0000001.pt --logdir {tmp_path}/out")
    lat = run(create_argparser(True).parse_args(flags.split()))
    assert lat.shape == (1, 12, 32, 32) and torch.isfinite(lat).all()
    meta = json.load(open(tmp_path / "out" / "args.json"))
    assert meta['conditioning'].startswith('MV23DConditioner(') and meta['weights']['dit'] == 'synthetic'
    # a different view set conditions differently
    np.savez(tmp_path / "views2.npz", img=synth_input('mvimg', (3, 3, 224, 224), 6).clamp(-1, 1).numpy(), c=cams)
    lat2 = run(create_argparser(True).parse_args(flags.replace("views.npz", "views2.npz").replace("/out", "/out2").split()))
    assert not torch.equal(lat, lat2)
