"""Loading released LN3Diff weights into the HIP-path modules.

The module mirrors keep the reference's state-dict keys (incl. the xformers FusedMLP naming `mlp.mlp.{0,2}.weight`,
`mlp.mlp.{1,3}.bias`), so loading is prefix resolution + a strict shape check:
  * diffusion model: saved by TrainLoop as `model_rec*/model_joint_denoise*` files or inside the HF `yslan/LN3Diff`
    safetensors under `ddpm_model.` (nsr/train_util_diffusion.py:780-843 loads with strict=True after stripping),
  * VAE decoder + tri-plane renderer: under `rec_model.decoder.` / `decoder.` (nsr/train_nv_util.py).
Files: `.safetensors` (safetensors.torch.load_file) or torch pickles of a flat state dict (optionally under 'state_dict').
"""
import torch

from . import _cache


def read_state_dict(path):
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(str(path))
    sd = torch.load(str(path), map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and 'state_dict' in sd and all(not torch.is_tensor(v) for v in sd.values() if v is not sd['state_dict']):
        sd = sd['state_dict']
    return sd


def load_into(module, sd, prefixes=("",), strict=True):
    """Copy every tensor of `module.state_dict()` from `sd`, trying the key prefixes in order.  Returns the prefix histogram.
    strict: missing keys or shape mismatches raise (like the reference's load_state_dict(strict=True))."""
    own = module.state_dict()
    out, used, bad = {}, {}, []
    for k, v in own.items():
        for p in prefixes:
            t = sd.get(p + k)
            if t is None:
                continue
            if tuple(t.shape) != tuple(v.shape):
                bad.append(f"{p + k}: checkpoint {tuple(t.shape)} vs model {tuple(v.shape)}")
                continue
            out[k] = t
            used[p] = used.get(p, 0) + 1
            break
    missing = [k for k in own if k not in out]
    if strict and (missing or bad):
        raise RuntimeError(f"checkpoint does not match the model: {len(missing)} missing (e.g. {missing[:4]}), "
                           f"{len(bad)} shape mismatches (e.g. {bad[:2]})")
    module.load_state_dict(out, strict=strict and not missing)
    _cache.bump()
    return used


DIT_PREFIXES = ("ddpm_model.", "module.", "model.", "")
DECODER_PREFIXES = ("rec_model.decoder.", "auto_encoder.decoder.", "decoder.", "module.decoder.", "")
CONDITIONER_PREFIXES = ("conditioner.embedders.0.", "cond_stage_model.", "")


def covers(sd, module, prefixes):
    """How many of the module's tensors the file holds under any of the prefixes (0: the file does not contain this component)."""
    own = module.state_dict()
    return sum(1 for k in own if any((p + k) in sd for p in prefixes))


def load_checkpoint(path, dit=None, decoder=None, conditioner=None, strict=True, skip_absent=False):
    """skip_absent: a component none of whose tensors is in the file is left alone (reported as absent) instead of raising - the
    joint `--resume_checkpoint` files hold denoiser AND decoder, the `--ddpm_model_path` / `--rec_model_path` ones only one."""
    sd = read_state_dict(path)
    rep = {}
    if skip_absent:
        if dit is not None and covers(sd, dit, DIT_PREFIXES) == 0:
            rep['dit'], dit = 'absent', None
        if decoder is not None and covers(sd, decoder, DECODER_PREFIXES) == 0:
            rep['decoder'], decoder = 'absent', None
    if dit is not None:
        rep['dit'] = load_into(dit, sd, DIT_PREFIXES, strict)
    if decoder is not None:
        rep['decoder'] = load_into(decoder, sd, DECODER_PREFIXES, strict)
    if conditioner is not None:
        rep['conditioner'] = load_into(conditioner, sd, CONDITIONER_PREFIXES, strict)
    return rep
