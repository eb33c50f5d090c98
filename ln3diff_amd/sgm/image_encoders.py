"""Image conditioners of the I23D path on the HIP kernels.

Mirrors sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder (/root/reference/sgm/modules/encoders/modules.py:578-733;
config sgm/configs/img23d-clipl-compat-fm-lognorm.yaml: arch 'ViT-L-14', version 'openai', output_tokens=True) and
FrozenDinov2ImageEmbedder (:735-869, dinov2_vitl14_reg, x_norm_patchtokens).  The towers themselves live in third-party
packages (open_clip, torch.hub facebookresearch/dinov2) that are not in the image; they are re-implemented here on the DiT's
kernels with the ORIGINAL packages' state-dict key layouts (`model.visual.*` / `model.*`), so released weights load without
renaming; the arithmetic is pinned against the architecture-identical HuggingFace models (tests/golden/make_golden_vit.py),
the key naming and the kornia resize of `preprocess` are not (inputs: 224x224, already resized; normalisation is done here).

Kernel sequence per block (same as the text tower): LayerNorm(+affine) -> fused QKV GEMM with head-split epilogue ->
attention kernel (257 / 261 tokens, Dh 64) -> out-proj GEMM with the residual epilogue (LayerScale gamma as its gate for
DINOv2) -> LayerNorm -> fc1 GEMM + quick-GELU / erf-GELU epilogue -> fc2 GEMM with the residual epilogue.  The 14x14
patch-embedding convolution is a patchify kernel + one GEMM (K = 588 padded to 640).
"""
import torch
import torch.nn as nn

from .. import ops, _cache
from ..dit.dit_models_xformers import Workspace, bf16, f32


# ----------------------------------------------------------------------------- parameter containers (original key layouts)
class _MHA(nn.Module):                       # open_clip / nn.MultiheadAttention naming
    def __init__(self, D):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.zeros(3 * D, D))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * D))
        self.out_proj = nn.Linear(D, D)


class _ClipMlp(nn.Module):
    def __init__(self, D, I):
        super().__init__()
        self.c_fc, self.c_proj = nn.Linear(D, I), nn.Linear(I, D)


class _ResBlock(nn.Module):
    def __init__(self, D, I):
        super().__init__()
        self.ln_1, self.attn, self.ln_2, self.mlp = nn.LayerNorm(D), _MHA(D), nn.LayerNorm(D), _ClipMlp(D, I)


class _ClipTransformer(nn.Module):
    def __init__(self, D, I, n):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResBlock(D, I) for _ in range(n)])


class _ClipVisual(nn.Module):
    def __init__(self, D, I, n, heads, S, P, proj):
        super().__init__()
        self.heads, self.image_size, self.patch = heads, S, P
        self.class_embedding = nn.Parameter(torch.zeros(D))
        self.positional_embedding = nn.Parameter(torch.zeros((S // P) ** 2 + 1, D))
        self.proj = nn.Parameter(torch.zeros(D, proj))
        self.conv1 = nn.Conv2d(3, D, P, P, bias=False)
        self.ln_pre, self.ln_post = nn.LayerNorm(D), nn.LayerNorm(D)
        self.transformer = _ClipTransformer(D, I, n)


class _ClipModel(nn.Module):                 # open_clip CLIP after `del model.transformer` (the text tower)
    def __init__(self, **kw):
        super().__init__()
        self.visual = _ClipVisual(**kw)


class _DinoAttn(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.qkv, self.proj = nn.Linear(D, 3 * D), nn.Linear(D, D)


class _Gamma(nn.Module):
    def __init__(self, D):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(D))


class _DinoMlp(nn.Module):
    def __init__(self, D, I):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(D, I), nn.Linear(I, D)


class _DinoBlock(nn.Module):
    def __init__(self, D, I):
        super().__init__()
        self.norm1, self.attn, self.ls1 = nn.LayerNorm(D, eps=1e-6), _DinoAttn(D), _Gamma(D)
        self.norm2, self.mlp, self.ls2 = nn.LayerNorm(D, eps=1e-6), _DinoMlp(D, I), _Gamma(D)


class _DinoPatch(nn.Module):
    def __init__(self, D, P, in_chans=3):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, D, P, P)


class _DinoModel(nn.Module):                 # dinov2 DinoVisionTransformer key layout
    def __init__(self, D, n, heads, S, P, R, ratio=4, in_chans=3):
        super().__init__()
        self.heads, self.image_size, self.patch = heads, S, P
        self.cls_token = nn.Parameter(torch.zeros(1, 1, D))
        self.pos_embed = nn.Parameter(torch.zeros(1, (S // P) ** 2 + 1, D))
        self.register_tokens = nn.Parameter(torch.zeros(1, R, D))
        self.mask_token = nn.Parameter(torch.zeros(1, D))
        self.patch_embed = _DinoPatch(D, P, in_chans)
        self.blocks = nn.ModuleList([_DinoBlock(D, ratio * D) for _ in range(n)])
        self.norm = nn.LayerNorm(D, eps=1e-6)


# ----------------------------------------------------------------------------- shared HIP runner
class _ViTRunner:
    """Packed weights + the kernel sequence of a pre-LN ViT; `spec` abstracts the two key layouts."""

    def __init__(self, spec, dev):
        self.s, self.dev, self.ws = spec, dev, Workspace(dev)
        D = spec['D']
        self.zeros = torch.zeros(D, device=dev)
        self.chans = spec['patch_w'].shape[1]
        kk = self.chans * spec['patch'] ** 2
        self.kpad = (kk + 63) // 64 * 64
        w = torch.zeros(D, self.kpad)
        w[:, :kk] = spec['patch_w'].detach().reshape(D, kk)
        self.pw, self.pb = bf16(w, dev), (f32(spec['patch_b'], dev) if spec['patch_b'] is not None else None)
        self.cls, self.pos = f32(spec['cls'].reshape(-1), dev), f32(spec['pos'].reshape(-1, D), dev)
        self.reg = f32(spec['reg'].reshape(-1, D), dev) if spec['reg'] is not None else None
        self.pre = tuple(f32(t, dev) for t in spec['pre_ln']) if spec['pre_ln'] is not None else None
        self.post = tuple(f32(t, dev) for t in spec['post_ln'])
        self.layers = []
        for l in spec['layers']:
            q = {k: (bf16(v, dev) if k.endswith('_w') and v.dim() == 2 else f32(v, dev)) for k, v in l.items() if v is not None}
            self.layers.append(q)

    @torch.no_grad()
    def __call__(self, img):
        s, ws, dev = self.s, self.ws, self.dev
        B, S, P, D, H, R = img.shape[0], s['size'], s['patch'], s['D'], s['heads'], (self.reg.shape[0] if self.reg is not None else 0)
        assert tuple(img.shape[1:]) == (self.chans, S, S), f"expects {self.chans} x {S} x {S} inputs (resize first)"
        G = S // P
        Lp, T = G * G, 1 + R + G * G
        M, tpad, Dh = B * T, (T + 63) // 64 * 64, D // H
        assert Dh in (64, 128)
        pm = ws.get('pm', (B * Lp, self.kpad), torch.bfloat16)
        ops.vit_patchify(img.contiguous().float(), pm, B, S, P, self.kpad, self.chans)
        pe = ws.get('pe', (B * Lp, D), torch.float32)
        ops.gemm(pm, self.pw, self.pb, ops.EPI_F32, pe)
        x = ws.get('x', (M, D), torch.float32)
        ops.vit_assemble(pe, self.cls, self.reg, self.pos, x, B, Lp, R, D)
        if self.pre is not None:
            ops.layernorm_f32(x, self.pre[0], self.pre[1], x, M, D, s['eps'])
        h = ws.get('h', (M, D), torch.bfloat16)
        q = ws.get('q', (B, H, tpad, Dh), torch.bfloat16, zero=True)
        k = ws.get('k', (B, H, tpad, Dh), torch.bfloat16, zero=True)
        vt = ws.get('vt', (B, H, Dh, tpad), torch.bfloat16, zero=True)
        o = ws.get('o', (M, D), torch.bfloat16)
        f1 = ws.get('f1', (M, self.layers[0]['fc1_w'].shape[0]), torch.bfloat16)
        act = ops.EPI_QUICK_GELU if s['act'] == 'quick_gelu' else ops.EPI_GELU_ERF
        for L in self.layers:
            ops.norm_modulate(x, h, M, D, kind=0, eps=s['eps'], weight=L['n1_w'], shift=L['n1_b'], scale=self.zeros, mod_rows=M, mod_ld=0)
            ops.gemm(h, L['qkv_w'], L['qkv_b'], ops.EPI_HEADS, q, k, vt, M=M, tokens=T, tok_pad=tpad, heads=H, head_dim=Dh,
                     transpose_mask=0b100)
            ops.attention(q, k, vt, o, B, H, T, tpad, T, tpad, Dh, scale=Dh ** -0.5)
            g1, g2 = L.get('ls1'), L.get('ls2')                       # LayerScale = a per-feature gate on the branch output
            ops.gemm(o, L['o_w'], L['o_b'], ops.EPI_GATE_RES, x, gate=g1, gate_rows=M, gate_ld=0)
            ops.norm_modulate(x, h, M, D, kind=0, eps=s['eps'], weight=L['n2_w'], shift=L['n2_b'], scale=self.zeros, mod_rows=M, mod_ld=0)
            ops.gemm(h, L['fc1_w'], L['fc1_b'], act, f1)
            ops.gemm(f1, L['fc2_w'], L['fc2_b'], ops.EPI_GATE_RES, x, gate=g2, gate_rows=M, gate_ld=0)
        y = torch.empty(B, T, D, device=dev, dtype=torch.float32)
        ops.layernorm_f32(x, self.post[0], self.post[1], y, M, D, s['eps'])
        return y, R


class _ImageEmbedderBase(nn.Module):
    MEAN, STD = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)

    def __init__(self, device="cuda", freeze=True):
        super().__init__()
        self.device = device
        self._runner = None
        self._epoch = -1
        _cache.watch(self)
        self.register_buffer("mean", torch.tensor(self.MEAN), persistent=False)
        self.register_buffer("std", torch.tensor(self.STD), persistent=False)

    def freeze(self):
        self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def preprocess(self, x):
        """[-1, 1] images -> the tower's input size -> normalised (sgm/modules/encoders/modules.py:633-645,802-814): kornia
        `geometry.resize(x, (S, S), 'bicubic', align_corners=True, antialias=self.antialias)`, then (x + 1) / 2 and mean / std."""
        S = self._spec_model().image_size
        # one HIP call (ln3d_image_preprocess): Gaussian pre-blur when a side shrinks, bicubic align_corners resize (identity taps
        # when the input is at size already), (x + 1) / 2, mean / std.  kornia is absent from this image: its published algorithm
        # is restated (oracle/vit_image.py::resize_bicubic_antialias is the CPU restatement the kernel is tested against) - PARITY
        # UNPINNED against kornia itself.
        return ops.image_preprocess(x, S, getattr(self, 'antialias', True), self.MEAN, self.STD)

    def _run(self, image, extra=None):
        if not image.is_cuda:
            raise RuntimeError("ln3diff_amd image embedders run on the HIP device only (no CPU fallback)")
        if image.dim() == 5:
            image = image.reshape(-1, *image.shape[2:])
        if self._runner is None or self._runner.dev != image.device or self._epoch != _cache.EPOCH[0]:
            self._runner = _ViTRunner(self._spec(), image.device)
            self._proj_bf = None
            self._epoch = _cache.EPOCH[0]
            _cache.watch_tree(self)
        x = self.preprocess(image.float())
        if extra is not None:                             # channels that bypass the image normalisation (Pluecker ray maps)
            x = torch.cat([x, extra.to(x.dtype)], 1)
        return self._runner(x)


class FrozenOpenCLIPImageEmbedder(_ImageEmbedderBase):
    """forward(image in [-1, 1]) -> (tokens [B, 256, 1024] = ln_post'ed patch tokens, pooled [B, 768]) with output_tokens=True,
    else pooled (reference forward :652-697; ucg dropout is a training-time feature and is not applied)."""
    MEAN, STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, arch="ViT-L-14", version="openai", device="cuda", max_length=77, freeze=True, antialias=True, ucg_rate=0.0,
                 unsqueeze_dim=False, repeat_to_max_len=False, num_image_crops=0, output_tokens=False, init_device=None,
                 width=1024, mlp_width=4096, layers=24, heads=16, image_size=224, patch_size=14, embed_dim=768):
        super().__init__(device, freeze)
        assert arch == "ViT-L-14" or width != 1024, "only the ViT-L/14 geometry has defaults; pass the dimensions explicitly"
        assert num_image_crops == 0 and not repeat_to_max_len
        self.model = _ClipModel(D=width, I=mlp_width, n=layers, heads=heads, S=image_size, P=patch_size, proj=embed_dim)
        self.output_tokens, self.unsqueeze_dim = output_tokens, unsqueeze_dim
        self.act = 'quick_gelu' if version == 'openai' else 'gelu'      # open_clip: QuickGELU only for the OpenAI weights
        if freeze:
            self.freeze()

    def _spec_model(self):
        return self.model.visual

    def _spec(self):
        v = self.model.visual
        D = v.class_embedding.shape[0]
        layers = []
        for b in v.transformer.resblocks:
            layers.append({'n1_w': b.ln_1.weight, 'n1_b': b.ln_1.bias, 'n2_w': b.ln_2.weight, 'n2_b': b.ln_2.bias,
                           'qkv_w': b.attn.in_proj_weight, 'qkv_b': b.attn.in_proj_bias, 'o_w': b.attn.out_proj.weight,
                           'o_b': b.attn.out_proj.bias, 'fc1_w': b.mlp.c_fc.weight, 'fc1_b': b.mlp.c_fc.bias,
                           'fc2_w': b.mlp.c_proj.weight, 'fc2_b': b.mlp.c_proj.bias, 'ls1': None, 'ls2': None})
        return {'D': D, 'heads': v.heads, 'size': v.image_size, 'patch': v.patch, 'eps': 1e-5, 'act': self.act,
                'patch_w': v.conv1.weight, 'patch_b': None, 'cls': v.class_embedding, 'pos': v.positional_embedding, 'reg': None,
                'pre_ln': (v.ln_pre.weight, v.ln_pre.bias), 'post_ln': (v.ln_post.weight, v.ln_post.bias), 'layers': layers}

    @torch.no_grad()
    def forward(self, image, no_dropout=False):
        y, _ = self._run(image)
        v = self.model.visual
        B, D = y.shape[0], y.shape[2]
        cls_bf = y[:, 0].to(torch.bfloat16).contiguous()
        pw = getattr(self, '_proj_bf', None)
        if pw is None or pw.device != y.device:
            pw = self._proj_bf = bf16(v.proj.t(), y.device)              # [embed_dim, width]
        z = torch.empty(B, pw.shape[0], device=y.device, dtype=torch.float32)
        ops.gemm(cls_bf, pw, None, ops.EPI_F32, z)
        z = z.to(image.dtype)
        if self.unsqueeze_dim:
            z = z[:, None, :]
        if self.output_tokens:
            return y[:, 1:].to(image.dtype), z
        return z

    def encode(self, image):
        return self(image)


class FrozenDinov2ImageEmbedder(_ImageEmbedderBase):
    """forward(image in [-1, 1]) -> x_norm_patchtokens [B, 256, 1024]; with output_cls=True -> (x_norm_clstoken, patch tokens)
    (reference :824-869)."""
    MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def __init__(self, arch="vitl", version="dinov2", device="cuda", max_length=77, freeze=True, antialias=True, ucg_rate=0.0,
                 unsqueeze_dim=False, repeat_to_max_len=False, num_image_crops=0, output_tokens=False, output_cls=False,
                 init_device=None, width=1024, layers=24, heads=16, image_size=224, patch_size=14, num_register_tokens=4):
        super().__init__(device, freeze)
        self.model = _DinoModel(D=width, n=layers, heads=heads, S=image_size, P=patch_size, R=num_register_tokens)
        self.output_cls = output_cls
        if freeze:
            self.freeze()

    def _spec_model(self):
        return self.model

    def _spec(self):
        m = self.model
        layers = []
        for b in m.blocks:
            layers.append({'n1_w': b.norm1.weight, 'n1_b': b.norm1.bias, 'n2_w': b.norm2.weight, 'n2_b': b.norm2.bias,
                           'qkv_w': b.attn.qkv.weight, 'qkv_b': b.attn.qkv.bias, 'o_w': b.attn.proj.weight, 'o_b': b.attn.proj.bias,
                           'fc1_w': b.mlp.fc1.weight, 'fc1_b': b.mlp.fc1.bias, 'fc2_w': b.mlp.fc2.weight, 'fc2_b': b.mlp.fc2.bias,
                           'ls1': b.ls1.gamma, 'ls2': b.ls2.gamma})
        return {'D': m.cls_token.shape[-1], 'heads': m.heads, 'size': m.image_size, 'patch': m.patch, 'eps': 1e-6, 'act': 'gelu',
                'patch_w': m.patch_embed.proj.weight, 'patch_b': m.patch_embed.proj.bias, 'cls': m.cls_token, 'pos': m.pos_embed,
                'reg': m.register_tokens if m.register_tokens.shape[1] > 0 else None, 'pre_ln': None,
                'post_ln': (m.norm.weight, m.norm.bias), 'layers': layers}

    @torch.no_grad()
    def forward(self, image, no_dropout=False, **kwargs):
        y, R = self._run(image)
        tokens = y[:, 1 + R:]
        if self.output_cls:
            return y[:, 0], tokens
        return tokens

    def encode(self, image):
        return self(image)


class FrozenDinov2ImageEmbedderMVPlucker(FrozenDinov2ImageEmbedder):
    """Multi-view conditioner of the released MV configs (sgm/modules/encoders/modules.py:871-1111; configs
    sgm/configs/mv23d-plucker-clipl-compat-fm-lognorm{,-noclip}.yaml: arch vitb, n_cond_frames 4): DINOv2-reg whose patch embedding
    takes 9 channels - the normalised RGB view followed by its 6 Pluecker ray maps (o x d, d) at the tower's 224 x 224 grid, built
    from the view's camera (16 c2w + 9 normalised intrinsics).  forward({'img': [B, T, 3, H, W] in [-1, 1], 'c': [B, T, 25]}) ->
    x_norm_patchtokens of the FIRST n_cond_frames views, [B, n_cond_frames, 256, D] = the MV denoisers' context['concat'].
    Training-time augmentation (ucg dropout, scale jitter, random camera rotation: aug_c) is not part of sampling and is not built.
    State-dict keys are the reference embedder's (`model.patch_embed.proj.weight` is [D, 9, 14, 14])."""
    ARCHS = {'vits': (384, 12, 6), 'vitb': (768, 12, 12), 'vitl': (1024, 24, 16)}

    def __init__(self, arch="vitb", version="dinov2", device="cuda", freeze=True, antialias=True, ucg_rate=0.0, output_tokens=False,
                 output_cls=False, init_device=None, n_cond_frames=4, enable_bf16=False, modLN=False, aug_c=False,
                 width=None, layers=None, heads=None, image_size=224, patch_size=14, num_register_tokens=4, **ignored):
        D, n, H = self.ARCHS[arch]
        super().__init__(arch=arch, version=version, device=device, freeze=False, antialias=antialias, output_cls=False,
                         width=width or D, layers=layers or n, heads=heads or H, image_size=image_size, patch_size=patch_size,
                         num_register_tokens=num_register_tokens)
        assert not output_cls and not aug_c
        self.n_cond_frames = n_cond_frames
        m = self.model
        self.model = _DinoModel(D=m.cls_token.shape[-1], n=len(m.blocks), heads=m.heads, S=m.image_size, P=m.patch,
                                R=m.register_tokens.shape[1], in_chans=9)
        if freeze:
            self.freeze()

    def get_plucker_ray(self, c):
        """c [V, 25] -> [V, 6, S, S] (reference :995-1005; one HIP call instead of a Python loop over views)."""
        return ops.plucker_rays(c, self.model.image_size)

    @torch.no_grad()
    def forward(self, img_c, no_dropout=False):
        img, c = img_c['img'], img_c['c']
        if img.dim() != 5 or c.dim() != 3 or c.shape[-1] != 25 or img.shape[1] < self.n_cond_frames or c.shape[:2] != img.shape[:2]:
            raise ValueError(f"expects {{'img': [B, T >= {self.n_cond_frames}, 3, H, W], 'c': [B, T, 25]}}; got {tuple(img.shape)} / {tuple(c.shape)}")
        B, T = img.shape[0], self.n_cond_frames
        views = img[:, :T].reshape(B * T, *img.shape[2:])
        rays = self.get_plucker_ray(c[:, :T].reshape(B * T, 25).to(img.device))
        y, R = self._run(views, extra=rays)
        tokens = y[:, 1 + R:]
        return tokens.reshape(B, T, tokens.shape[1], tokens.shape[2]).to(img.dtype)


class MV23DConditioner(nn.Module):
    """GeneralConditioner semantics of the multi-view configs: the MVPlucker embedder's 4-D output is routed to 'concat'
    ([B, V, 256, D], the MV denoisers' cross-attention context); with a CLIP embedder (the non-noClip config: the FIRST view through
    FrozenOpenCLIPImageMVEmbedder, :1663-1684) its tokens / pooled vector become 'crossattn' / 'vector'."""

    def __init__(self, dino_mv, clip=None):
        super().__init__()
        self.dino_mv, self.clip = dino_mv, clip
        assert clip is None or clip.output_tokens

    @torch.no_grad()
    def forward(self, img_c):
        out = {'concat': self.dino_mv(img_c)}
        if self.clip is not None:
            out['crossattn'], out['vector'] = self.clip(img_c['img'][:, 0])
        return out

    def get_unconditional_conditioning(self, cond):
        return cond, {k: torch.zeros_like(v) for k, v in cond.items()}


class I23DConditioner(nn.Module):
    """GeneralConditioner semantics for the I23D config (sgm/modules/encoders/modules.py:80-191 with
    sgm/configs/img23d-clipl-compat-fm-lognorm.yaml): embedder outputs are routed by rank (2-D -> 'vector', 3-D -> 'crossattn')
    and concatenated along the feature axis: crossattn = [CLIP tokens (1024) || DINOv2 patch tokens (1024)] = [B, 256, 2048],
    vector = CLIP pooled [B, 768] - the context dict DiT_I23D_PixelArt.forward consumes.  The unconditional branch of CFG is
    the zero embedding (ucg with zeroed outputs)."""

    def __init__(self, clip=None, dino=None):
        super().__init__()
        self.clip = clip if clip is not None else FrozenOpenCLIPImageEmbedder(output_tokens=True)
        self.dino = dino if dino is not None else FrozenDinov2ImageEmbedder()
        assert self.clip.output_tokens and not self.dino.output_cls

    @torch.no_grad()
    def forward(self, img):
        tokens, pooled = self.clip(img)
        dino = self.dino(img)
        return {'crossattn': torch.cat([tokens, dino], 2), 'vector': pooled}

    def get_unconditional_conditioning(self, cond):
        uc = {k: torch.zeros_like(v) for k, v in cond.items()}
        return cond, uc
