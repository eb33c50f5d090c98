"""Import harness for the read-only reference checkout (THIS container only).

Used by tests/golden/make_golden.py to (i) validate oracle/ against the reference's
own Python and (ii) emit the committed golden vectors.  Nothing here is imported
by the product, by `-m gpu` tests, by smoke() or by bench.py: /root/reference does
not exist on the GPU box.

The reference cannot be imported as-is: its package __init__ files pull in every
trainer/encoder, and several third-party packages it needs are not installed
(xformers, timm, torchdiffeq, omegaconf, ...).  We therefore
  * register empty package shells for the reference's own packages so that their
    __init__.py never runs (sub-modules still import from the real files);
  * stub import-time-only dependencies with MagicMock;
  * provide *functional* stand-ins for the third-party arithmetic that sits on
    the hot path (xformers attention / FusedMLP, timm PatchEmbed / Mlp,
    torchdiffeq fixed-grid odeint).  Those four are therefore "parity unpinned"
    against the real third-party code - they implement the
    published semantics: exact softmax attention, Linear-GELU(erf)-Linear,
    strided-conv patch embedding, explicit Euler / Heun on the supplied grid.
"""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = os.environ.get("LN3DIFF_REFERENCE", "/root/reference")


def _shell(name):
    """Empty package whose __path__ points at the reference directory."""
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, *name.split("."))]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _mock(name):
    m = MagicMock(name=name)
    m.__path__ = []
    m.__spec__ = None
    sys.modules[name] = m
    return m


# ---------------------------------------------------------------- timm shims
class _PatchEmbed(nn.Module):
    """timm 0.6 PatchEmbed semantics: Conv2d(k=s=patch) -> flatten(2).transpose(1,2)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768,
                 norm_layer=None, flatten=True, bias=True):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size,
                              stride=patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


class _Mlp(nn.Module):
    """timm Mlp: fc1 -> act -> fc2 (dropouts are p=0 on this path)."""

    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, bias=True, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer() if isinstance(act_layer, type) or callable(act_layer) and not isinstance(act_layer, nn.Module) else act_layer
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


# ------------------------------------------------------------- xformers shims
def _mea(q, k, v, attn_bias=None, op=None, p=0.0, scale=None):
    """xformers.ops.memory_efficient_attention: softmax(q k^T / sqrt(Dh)) v.
    Accepts [B, N, H, Dh] (4-D) and [B*H, N, Dh] (3-D) layouts."""
    assert attn_bias is None
    if q.ndim == 4:
        q_, k_, v_ = (t.permute(0, 2, 1, 3) for t in (q, k, v))
        o = F.scaled_dot_product_attention(q_, k_, v_, scale=scale)
        return o.permute(0, 2, 1, 3)
    return F.scaled_dot_product_attention(q, k, v, scale=scale)


class _FusedMLP(nn.Module):
    """xformers FusedMLP(dim, dropout=0, GeLU, mult): Linear(no bias) -> (+bias, erf-GELU)
    -> Linear(no bias) -> (+bias).  Parameter names follow xformers 0.0.26:
    mlp.0.weight, mlp.1.bias, mlp.2.weight, mlp.3.bias."""

    class _Bias(nn.Module):
        def __init__(self, n, act):
            super().__init__()
            self.bias = nn.Parameter(torch.zeros(n))
            self.act = act

        def forward(self, x):
            x = x + self.bias
            return F.gelu(x) if self.act else x

    def __init__(self, dim_model, dropout, activation, hidden_layer_multiplier, bias=True):
        super().__init__()
        h = hidden_layer_multiplier * dim_model
        self.mlp = nn.Sequential(
            nn.Linear(dim_model, h, bias=False), self._Bias(h, True),
            nn.Linear(h, dim_model, bias=False), self._Bias(dim_model, False))

    def forward(self, x):
        return self.mlp(x)


# ---------------------------------------------------------- torchdiffeq shim
def _odeint(fn, x, t, method="euler", atol=None, rtol=None, **kw):
    """Fixed-grid explicit solvers on the supplied time grid (torchdiffeq semantics for
    'euler' / 'heun'; its fixed-grid 'heun' is not offered upstream as such - 'midpoint'
    and 'rk4' follow the textbook tableaux).  Adaptive dopri5 is NOT reproduced."""
    assert method in ("euler", "heun", "midpoint", "rk4"), method
    ys = [x]
    for i in range(len(t) - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        y = ys[-1]
        if method == "euler":
            y = y + dt * fn(t0, y)
        elif method == "heun":
            k1 = fn(t0, y)
            k2 = fn(t1, y + dt * k1)
            y = y + dt * 0.5 * (k1 + k2)
        elif method == "midpoint":
            k1 = fn(t0, y)
            y = y + dt * fn(t0 + 0.5 * dt, y + 0.5 * dt * k1)
        else:
            k1 = fn(t0, y)
            k2 = fn(t0 + dt / 3, y + dt * k1 / 3)
            k3 = fn(t0 + dt * 2 / 3, y + dt * (k2 - k1 / 3))
            k4 = fn(t1, y + dt * (k1 - k2 + k3))
            y = y + dt * (k1 + 3 * (k2 + k3) + k4) / 8
        ys.append(y)
    return torch.stack(ys, 0)


_INSTALLED = False


def install():
    global _INSTALLED
    if _INSTALLED:
        return
    _INSTALLED = True
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    for pkg in ("nsr", "guided_diffusion", "sgm", "sgm.modules",
                "sgm.modules.diffusionmodules", "ldm", "ldm.modules",
                "ldm.modules.diffusionmodules", "nsr.lsgm", "nsr.losses",
                "sgm.modules.encoders", "sgm.modules.autoencoding"):
        _shell(pkg)

    for name in ("blobfile", "torchvision", "torchvision.transforms", "torchvision.utils",
                 "torchvision.models", "torchvision.transforms.functional",
                 "kornia", "lpips", "cv2", "imageio", "tensorboard",
                 "torch.utils.tensorboard", "torch.utils.tensorboard.writer",
                 "kiui", "kiui.op", "kiui.cam", "pytorch_lightning", "omegaconf", "open_clip",
                 "beartype", "beartype.typing", "beartype.door", "clip", "lmdb",
                 "webdataset", "mcubes", "trimesh", "skimage", "skimage.metrics",
                 "apex", "apex.normalization", "diffusers", "taming", "point_cloud_utils",
                 "pytorch3d", "open3d", "plyfile", "xatlas", "nvdiffrast", "nvdiffrast.torch",
                 "torch_scatter", "vision_aided_loss", "piq", "click", "ipdb", "mpi4py",
                 "matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image", "timm.layers",
                 "tqdm.contrib", "easydict", "gradio", "rembg", "pymeshlab", "pytorch_msssim",
                 "kaolin"):
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except Exception:
            _mock(name)
    # apex must FAIL to import so the reference takes its torch fallbacks
    for name in ("apex", "apex.normalization"):
        sys.modules[name] = None  # type: ignore

    # timm
    timm = types.ModuleType("timm"); timm.__path__ = []
    tm = types.ModuleType("timm.models"); tm.__path__ = []
    tv = types.ModuleType("timm.models.vision_transformer")
    tv.PatchEmbed, tv.Mlp, tv.Attention = _PatchEmbed, _Mlp, MagicMock()
    tl = types.ModuleType("timm.models.layers")
    tl.PatchEmbed, tl.Mlp = _PatchEmbed, _Mlp
    tl.trunc_normal_ = nn.init.trunc_normal_
    tl.DropPath = nn.Identity
    tl.to_2tuple = lambda x: (x, x) if not isinstance(x, tuple) else x
    sys.modules.update({"timm": timm, "timm.models": tm,
                        "timm.models.vision_transformer": tv, "timm.models.layers": tl})
    timm.models = tm; tm.vision_transformer = tv; tm.layers = tl

    # xformers
    xf = types.ModuleType("xformers"); xf.__path__ = []
    xo = types.ModuleType("xformers.ops")
    xo.memory_efficient_attention = _mea
    xo.unbind = torch.unbind
    xo.fmha = MagicMock()
    xo.MemoryEfficientAttentionFlashAttentionOp = None
    xc = types.ModuleType("xformers.components"); xc.__path__ = []
    xa = types.ModuleType("xformers.components.activations")

    class Activation:
        GeLU = "gelu"
    xa.Activation = Activation
    xa.build_activation = MagicMock()
    xff = types.ModuleType("xformers.components.feedforward"); xff.__path__ = []
    xfm = types.ModuleType("xformers.components.feedforward.fused_mlp")
    xfm.FusedMLP = _FusedMLP
    xff.fused_mlp = xfm
    sys.modules.update({"xformers": xf, "xformers.ops": xo, "xformers.components": xc,
                        "xformers.components.activations": xa,
                        "xformers.components.feedforward": xff,
                        "xformers.components.feedforward.fused_mlp": xfm})
    xf.ops = xo; xf.components = xc

    td = types.ModuleType("torchdiffeq")
    td.odeint = _odeint
    sys.modules["torchdiffeq"] = td


def ref_dit_modules():
    """dit.dit_models_xformers guards its FusedMLP import with torch.cuda.is_available();
    inject the names it would have imported (reference dit/dit_models_xformers.py:39-43)."""
    install()
    dmx = importlib.import_module("dit.dit_models_xformers")
    dmx.fused_mlp = sys.modules["xformers.components.feedforward.fused_mlp"]
    dmx.Activation = sys.modules["xformers.components.activations"].Activation
    return dmx
