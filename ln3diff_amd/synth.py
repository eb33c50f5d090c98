"""Deterministic synthetic weights / inputs for the sampling hot path.

There is no network for checkpoints, so tests, goldens and bench.py all use weights
generated on the CPU from (name, shape, seed).  The reference initialises several
tensors to exactly zero (adaLN-zero, final_layer.linear, cap_embedder[-1];
reference dit/dit_models_xformers.py:807-819, dit/dit_i23d.py:207-217) which would
make every denoiser output 0 and parity vacuous - so every tensor gets a small
random value here.  torch's CPU generator is bit-reproducible
for a given torch build, which is what lets a golden output computed in the build
container be compared on the GPU box without shipping the weights.
"""
import zlib

import torch


def _scale_rule(name, shape):
    """(offset, scale) such that tensor = offset + scale * N(0,1)."""
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if 'pos_embed' in name:
        raise ValueError('pos_embed is computed, not synthesised')
    if name.endswith('scale_shift_table'):
        return 0.0, 1.0 / shape[-1] ** 0.5
    if leaf == 'weight' and len(shape) == 1:                 # norm scales
        return 1.0, 0.05
    if leaf == 'bias' or len(shape) == 1:
        return 0.0, 0.02
    if 'triplane_decoder' in name or name.startswith('net.'):  # OSGDecoder FullyConnectedLayer: N(0,1)
        return 0.0, 1.0
    if 'adaLN_modulation' in name or 'final_layer.linear' in name:
        return 0.0, 0.02
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return 0.0, 0.7 / fan_in ** 0.5


def synth_tensor(name, shape, seed=0):
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    off, sc = _scale_rule(name, shape)
    r = torch.randn(tuple(shape), generator=g, dtype=torch.float32)
    if name.endswith('scale_shift_table'):
        return r / tuple(shape)[-1] ** 0.5          # bit-exact with the recipe the goldens were made with
    return r * sc + off if off else r * sc


def fill_module_random_(module, seed=0, device=None):
    """Fast variant for bench.py: same scale rules, device RNG, in place (not golden-reproducible)."""
    g = torch.Generator(device=device or 'cpu')
    g.manual_seed(seed)
    for k, v in module.state_dict().items():
        if 'pos_embed' in k or not v.dtype.is_floating_point:
            continue
        off, sc = _scale_rule(k, v.shape)
        v.copy_(torch.randn(v.shape, generator=g, device=g.device, dtype=torch.float32).mul_(sc).add_(off))
    from . import _cache
    _cache.bump()          # in-place parameter writes: packed device copies of the weights are stale now
    return module


def load_synth_(module, seed=0):
    """Fill a module with the deterministic synthetic weights the goldens were made with (synth_state_dict of its own key / shape
    manifest; computed positional embeddings are kept).  Returns (state_dict, shapes)."""
    sd = module.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    computed = {k: v for k, v in sd.items() if 'pos_embed' in k}
    new = synth_state_dict(shapes, seed, computed)
    module.load_state_dict(new, strict=True)
    from . import _cache
    _cache.bump()
    return new, shapes


def synth_state_dict(shapes, seed=0, computed=None):
    """shapes: {name: shape}.  computed: {name: tensor} for deterministic non-random
    entries (pos_embed)."""
    computed = computed or {}
    out = {}
    for k, shp in shapes.items():
        out[k] = computed[k].clone() if k in computed else synth_tensor(k, shp, seed)
    return out


def synth_input(name, shape, seed=0, scale=1.0):
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(('in:' + name).encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale


def orbit_cameras(n_views, radius=1.7719, elevation_deg=15.0, focal=1.3889):
    """[V,25] cameras on an orbit looking at the origin (OpenCV cam2world, normalised
    intrinsics fx=fy=focal, cx=cy=0.5) - the layout of the reference's
    assets/objv_eval_pose.pt (16 cam2world + 9 intrinsics)."""
    import math
    cams = []
    for i in range(n_views):
        az = 2 * math.pi * i / n_views
        el = math.radians(elevation_deg)
        pos = torch.tensor([radius * math.cos(el) * math.cos(az),
                            radius * math.cos(el) * math.sin(az),
                            radius * math.sin(el)])
        fwd = -pos / pos.norm()                               # camera +z looks at the origin
        up = torch.tensor([0., 0., 1.])
        right = torch.linalg.cross(fwd, up)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)                 # OpenCV: +y is down
        c2w = torch.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
        K = torch.tensor([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1.]])
        cams.append(torch.cat([c2w.reshape(-1), K.reshape(-1)]))
    return torch.stack(cams)


def synth_vit_state_dict(shapes, seed=0):
    """synth_state_dict for the image-conditioner towers: learned positional embeddings (0.02 * N) and LayerScale gammas
    (1 + 0.1 * N, so that the scaled branches matter in parity tests) get their own rules."""
    comp = {}
    for k, shp in shapes.items():
        if k.endswith('pos_embed'):
            comp[k] = synth_input('w:' + k, shp, seed, 0.02)
        elif k.endswith('.gamma'):
            comp[k] = 1.0 + synth_input('w:' + k, shp, seed, 0.1)
    return synth_state_dict(shapes, seed, comp)
