"""Entry-point flag surface and up-front validation (no GPU): scripts/vit_triplane_diffusion_sample{,_objaverse}.py -> ln3diff_amd.entry."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from ln3diff_amd.entry import TRAINERS, create_argparser, validate


def _args(objaverse=True, *flags):
    return create_argparser(objaverse).parse_known_args(list(flags))[0]


def test_reference_launcher_flags_parse():
    # the released T23D launcher's flags (shell_scripts/final_release/inference/sample_obajverse_t23d_dit.sh), booleans as words
    a, unknown = create_argparser(True).parse_known_args(
        "--dit_model_arch DiT-L/2 --trainer_name sgm_legacy --num_samples 4 --unconditional_guidance_scale 6.5 "
        "--triplane_scaling_divider 0.96806 --export_mesh True --use_amp False --lr 1e-4 --batch_size 4 --logdir /tmp/x "
        "--some_training_only_flag 3".split())
    assert a.dit_model_arch == 'DiT-L/2' and a.export_mesh is True and a.triplane_scaling_divider == 0.96806
    assert unknown == ['--some_training_only_flag', '3']
    assert validate(a) == 'edm'
    b = _args(False)
    assert b.trainer_name == 'adm' and b.triplane_scaling_divider == 1.0 and validate(b) == 'gd'


@pytest.mark.parametrize("flags,msg", [
    (("--dit_model_arch", "DiT-PixArt-L/2"), "I23D architecture"),                       # I23D arch without --i23d
    (("--i23d", "true", "--dit_model_arch", "DiT-PixelArt-L/2", "--trainer_name", "flow_matching"), "T23D architecture"),
    (("--i23d", "true", "--dit_model_arch", "DiT-PixArt-L/2"), "flow-matching"),         # I23D with the EDM engine
    (("--trainer_name", "no_such"), "known engines"),
    (("--trainer_name", "flow_matching"), "needs an I23D denoiser"),
    (("--dit_model_arch", "DiT-PixelArt-L/2"), "flow-matching T23D denoiser"),           # PixArt T23D with the EDM engine                     # T23D arch with the flow-matching engine
    (("--create_controlnet", "true"), "ControlNet"),
    (("--arch_dit_decoder", "DiT2-Z/9"), "arch_dit_decoder"),
    (("--num_samples", "0"), ">= 1"),
])
def test_unrunnable_flag_combinations_are_refused(flags, msg):
    with pytest.raises(SystemExit) as e:
        validate(_args(True, *flags))
    assert msg in str(e.value)


def test_registry_is_chosen_by_the_i23d_flag():
    """'DiT-L/2' names the text-conditioned DiT_TriLatent without --i23d and the plain image-conditioned DiT_I23D with it, like
    guided_diffusion/script_util.py picks DiT_models_i23d / DiT_models_t23d."""
    assert validate(_args(True, "--dit_model_arch", "DiT-L/2")) == 'edm'
    assert validate(_args(True, "--i23d", "true", "--dit_model_arch", "DiT-L/2", "--trainer_name", "flow_matching")) == 'flow'


@pytest.mark.parametrize("flags,msg", [
    (("--mixed_prediction", "True"), "only the U-Net denoiser"),
    (("--create_dit", "false"), "guided_diffusion engines"),
    (("--ae_classname", "vit.vit_triplane.SomethingElse"), "released decoder class"),
    (("--vae_p", "4"), "vae_p = 2"),
    (("--denoise_out_channels", "8"), "denoise_out_channels"),
    (("--out_chans", "48"), "3 planes x 32"),
    (("--i23d", "true", "--trainer_name", "flow_matching", "--dit_model_arch", "DiT-PixArt-MV-PCD-L"), "point-cloud denoiser"),
])
def test_model_describing_flags_are_checked_not_dropped(flags, msg):
    """Flags of the released launchers that describe the model / prediction type (they used to fall into the unused
    bucket): accepted at the released values, refused with the reason otherwise."""
    with pytest.raises(SystemExit) as e:
        validate(_args(True, *flags))
    assert msg in str(e.value)


def test_released_launcher_values_of_checked_flags_pass():
    # sample_obajverse_t23d_dit.sh: --pred_type v --predict_v True --mixed_prediction False --patch_size 14 --vae_p 2 ... with sgm_legacy
    a = _args(True, "--pred_type", "v", "--predict_v", "True", "--mixed_prediction", "False", "--patch_size", "14", "--vae_p", "2",
              "--denoise_out_channels", "4", "--decoder_in_chans", "32", "--out_chans", "96", "--triplane_in_chans", "32", "--decoder_output_dim", "3",
              "--ae_classname", "vit.vit_triplane.RodinSR_256_fusionv6_ConvQuant_liteSR_dinoInit3DAttn_SD_B_3L_C_withrollout_withSD_D_ditDecoder")
    assert validate(a) == 'edm'                     # the sgm engine never reads predict_v / pred_type (the reference neither)
    with pytest.raises(SystemExit) as e:            # ... the guided_diffusion engines do: ModelMeanType.V belongs to the U-Net denoiser
        validate(_args(False, "--create_dit", "true", "--predict_v", "True"))
    assert "epsilon only" in str(e.value)
    # sample_shapenet_*_t23d.sh: the U-Net, v-prediction with mixed prediction, DDIM 250
    shapenet = ("--create_dit", "false", "--trainer_name", "vpsde_crossattn", "--num_channels", "320", "--num_res_blocks", "2", "--num_heads", "8",
                "--attention_resolutions", "4,2,1", "--use_spatial_transformer", "True", "--transformer_depth", "1", "--context_dim", "768",
                "--denoise_in_channels", "12", "--denoise_out_channels", "12", "--roll_out", "false", "--predict_v", "True", "--pred_type", "v",
                "--mixed_prediction", "True", "--use_ddim", "True", "--timestep_respacing", "ddim250")
    assert validate(_args(False, *shapenet)) == 'gd'
    # r6: the second script's defaults are the reference's (create_dit False, roll_out False), so the launcher's own flag set -
    # which passes neither - selects the U-Net with the 12-channel latent
    bare = tuple(f for i, f in enumerate(shapenet) if f not in ("--create_dit", "--roll_out") and shapenet[i - 1] not in ("--create_dit", "--roll_out"))
    a = _args(False, *bare)
    assert a.create_dit is False and a.roll_out is False and validate(a) == 'gd'
    assert _args(True).create_dit is True and _args(True).roll_out is True
    # r6: the second script's defaults are the reference's (create_dit False, roll_out False), so the launcher's own flag set -
    # which passes neither - selects the U-Net with the 12-channel latent
    bare = tuple(f for i, f in enumerate(shapenet) if f not in ("--create_dit", "--roll_out") and shapenet[i - 1] not in ("--create_dit", "--roll_out"))
    a = _args(False, *bare)
    assert a.create_dit is False and a.roll_out is False and validate(a) == 'gd'
    assert _args(True).create_dit is True and _args(True).roll_out is True
    with pytest.raises(SystemExit) as e:            # v-prediction without the mixing branch: the reference's p_mean_variance asserts
        validate(_args(False, *[("False" if f == "True" and shapenet[i - 1] == "--mixed_prediction" else f) for i, f in enumerate(shapenet)]))
    assert "needs --mixed_prediction" in str(e.value)


def test_conditioning_is_never_silently_synthetic():
    """a prompt / image that cannot be encoded is refused; synthetic conditioning is labelled as such."""
    from ln3diff_amd.entry import load_conditioning
    c, src = load_conditioning(_args(True), 'cpu', True)
    assert src == 'synthetic' and c['crossattn'].shape == (1, 77, 768)
    c, src = load_conditioning(_args(False), 'cpu', False)              # the second script's DEFAULT prompt: allowed, labelled
    assert src == 'synthetic'
    with pytest.raises(SystemExit) as e:          # an EXPLICIT prompt equal to the reference script's default is still a prompt
        load_conditioning(_args(False, "--prompt", "a red chair"), 'cpu', False)
    assert "--clip_checkpoint" in str(e.value)
    for flags, msg in ((("--prompt", "a blue car"), "--clip_checkpoint"),
                       (("--i23d", "true", "--image_path", "/tmp/x.npy"), "--clip_checkpoint"),
                       (("--image_path", "/tmp/x.npy"), "--i23d"),
                       (("--i23d", "true", "--prompt", "a blue car"), "T23D")):
        with pytest.raises(SystemExit) as e:
            load_conditioning(_args(True, *flags), 'cpu', True)
        assert msg in str(e.value), (flags, str(e.value))


def test_trainer_table_covers_the_reference_names():
    for name in ('sgm_legacy', 'flow_matching', 'adm', 'vpsde_crossattn'):
        assert name in TRAINERS


@pytest.mark.parametrize("script", ["vit_triplane_diffusion_sample_objaverse.py", "vit_triplane_diffusion_sample.py"])
def test_scripts_fail_loudly_without_gpu(script):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), "--logdir", "/tmp/ln3d_entry_test"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.parametrize("argv,env,msg", [
    (["--gpus", "2"], {}, "exposes"),                                  # self-spawn refuses when the box has fewer GPUs
    (["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, "launcher started 2"),   # launcher / flag mismatch
    (["--gpus", "2"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, "launcher started 4"),
])
def test_bench_refuses_wrong_gpu_counts(argv, env, msg):
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU box")
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode != 0 and msg in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])
    assert '"metric"' not in r.stdout


def test_renderer_variants_outside_the_hot_path_are_refused():
    """Triplane(sr_kwargs=..., bcg_synthesis_kwargs=..., lrm_decoder=True) change what the ray marcher composites
    (nsr/triplane.py:476-500): not built, and not silently ignored either.  No released sampler builds them (every launcher passes
    --sr_training False -> sr_kwargs = {}).  decoder_output_dim = 32 (the ShapeNet launchers) IS taken: without the SR module only its
    first 3 colour rows reach image_raw, and the state dict keeps the reference's 33-row shape."""
    from ln3diff_amd.nsr.triplane import Triplane
    assert Triplane(img_resolution=16, sr_kwargs={}, bcg_synthesis_kwargs={}).superresolution is None
    for kw in (dict(sr_kwargs={'channel_base': 32768}), dict(lrm_decoder=True), dict(decoder_output_dim=2), dict(bcg_synthesis_kwargs={'a': 1})):
        with pytest.raises(NotImplementedError):
            Triplane(img_resolution=16, **kw)
    tp = Triplane(img_resolution=16, decoder_output_dim=32)
    assert tuple(tp.state_dict()['decoder.net.2.weight'].shape) == (33, 64) and tp.superresolution is None


def test_encoded_images_are_read_through_pillow(tmp_path):
    """--image_path accepts what Pillow decodes (RGBA composited on white, as the released demo inputs are).  Video encoding is outside
    the metric and the hot path: frames leave as .npy arrays + a .ppm per sample."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    import numpy as np
    import torch
    from ln3diff_amd.entry import _read_image
    rgba = np.zeros((8, 6, 4), np.uint8)
    rgba[..., 0] = 255                     # red ...
    rgba[:4, :, 3] = 255                   # ... opaque in the top half, transparent below
    p = str(tmp_path / "in.png")
    Image.fromarray(rgba, "RGBA").save(p)
    a = _read_image(p)
    assert a.shape == (1, 3, 8, 6) and float(a.min()) >= -1.0 and float(a.max()) <= 1.0
    assert torch.allclose(a[0, :, 0, 0], torch.tensor([1.0, -1.0, -1.0]))            # opaque red
    assert torch.allclose(a[0, :, 7, 0], torch.tensor([1.0, 1.0, 1.0]))              # transparent -> white background
