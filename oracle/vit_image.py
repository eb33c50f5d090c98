"""CPU restatement (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py) of the image conditioners on the I23D path.

sgm.modules.encoders.modules.FrozenOpenCLIPImageEmbedder (/root/reference/sgm/modules/encoders/modules.py:578-733, arch
ViT-L-14 'openai', output_tokens=True) wraps open_clip's VisionTransformer and FrozenDinov2ImageEmbedder (:735-869) wraps the
torch.hub DINOv2 ViT-L/14 with registers; both packages are third-party and absent from the reference tree AND from this image,
so the algorithms are restated from their published definitions with the packages' own state-dict key layouts, and the
arithmetic is pinned against the architecture-identical HuggingFace models that ARE installed here
(CLIPVisionModelWithProjection / Dinov2WithRegistersModel, tests/golden/make_golden_vit.py).  Key naming and the kornia
bicubic+antialias resize of `preprocess` stay unpinned (inputs here are already 224x224 and normalised).

open_clip VisionTransformer (output_tokens=True, final_ln_after_pool=False): conv1 (no bias) -> [class_embedding ; patches] +
positional_embedding -> ln_pre -> resblocks (pre-LN, fused in_proj, quick-GELU for the 'openai' weights) -> ln_post on ALL
tokens -> pooled = x[:, 0] @ proj, tokens = x[:, 1:].
DINOv2 (forward_features, is_training=True): patch_embed (bias) ; [cls ; patches] + pos_embed ; registers inserted after cls
-> blocks (pre-LN, fused qkv, LayerScale ls1/ls2, erf-GELU) -> norm -> x_norm_clstoken = x[:, 0], x_norm_patchtokens =
x[:, 1 + R:].
"""
import torch
import torch.nn.functional as F


def _attn(x, w_qkv, b_qkv, w_o, b_o, heads):
    B, T, D = x.shape
    q, k, v = F.linear(x, w_qkv, b_qkv).chunk(3, -1)
    q, k, v = (t.reshape(B, T, heads, D // heads).transpose(1, 2) for t in (q, k, v))
    a = torch.softmax(q @ k.transpose(-1, -2) * (D // heads) ** -0.5, -1) @ v
    return F.linear(a.transpose(1, 2).reshape(B, T, D), w_o, b_o)


def _patches(img, w, b, p):
    x = F.conv2d(img, w, b, stride=p)                 # [B, D, G, G]
    return x.flatten(2).transpose(1, 2)               # [B, G*G, D]


def openclip_visual_forward(sd, img, heads, patch=14, eps=1e-5, prefix='visual.'):
    g = lambda k: sd[prefix + k].float()
    x = _patches(img.float(), g('conv1.weight'), None, patch)
    B, _, D = x.shape
    x = torch.cat([g('class_embedding').expand(B, 1, D), x], 1) + g('positional_embedding')[None]
    x = F.layer_norm(x, (D,), g('ln_pre.weight'), g('ln_pre.bias'), eps)
    n = 1 + max(int(k[len(prefix):].split('.')[2]) for k in sd if k.startswith(prefix + 'transformer.resblocks.'))
    for i in range(n):
        L = f'transformer.resblocks.{i}.'
        h = F.layer_norm(x, (D,), g(L + 'ln_1.weight'), g(L + 'ln_1.bias'), eps)
        x = x + _attn(h, g(L + 'attn.in_proj_weight'), g(L + 'attn.in_proj_bias'), g(L + 'attn.out_proj.weight'),
                      g(L + 'attn.out_proj.bias'), heads)
        h = F.layer_norm(x, (D,), g(L + 'ln_2.weight'), g(L + 'ln_2.bias'), eps)
        h = F.linear(h, g(L + 'mlp.c_fc.weight'), g(L + 'mlp.c_fc.bias'))
        h = h * torch.sigmoid(1.702 * h)
        x = x + F.linear(h, g(L + 'mlp.c_proj.weight'), g(L + 'mlp.c_proj.bias'))
    pre_ln = x
    x = F.layer_norm(x, (D,), g('ln_post.weight'), g('ln_post.bias'), eps)
    pooled = x[:, 0] @ g('proj')
    return pooled, x[:, 1:], pre_ln


def dinov2_forward(sd, img, heads, patch=14, eps=1e-6):
    g = lambda k: sd[k].float()
    x = _patches(img.float(), g('patch_embed.proj.weight'), g('patch_embed.proj.bias'), patch)
    B, _, D = x.shape
    x = torch.cat([g('cls_token').expand(B, 1, D), x], 1) + g('pos_embed')
    R = sd['register_tokens'].shape[1] if 'register_tokens' in sd else 0
    if R:
        x = torch.cat([x[:, :1], g('register_tokens').expand(B, R, D), x[:, 1:]], 1)
    n = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('blocks.'))
    for i in range(n):
        L = f'blocks.{i}.'
        h = F.layer_norm(x, (D,), g(L + 'norm1.weight'), g(L + 'norm1.bias'), eps)
        x = x + g(L + 'ls1.gamma') * _attn(h, g(L + 'attn.qkv.weight'), g(L + 'attn.qkv.bias'), g(L + 'attn.proj.weight'),
                                            g(L + 'attn.proj.bias'), heads)
        h = F.layer_norm(x, (D,), g(L + 'norm2.weight'), g(L + 'norm2.bias'), eps)
        h = F.gelu(F.linear(h, g(L + 'mlp.fc1.weight'), g(L + 'mlp.fc1.bias')))
        x = x + g(L + 'ls2.gamma') * F.linear(h, g(L + 'mlp.fc2.weight'), g(L + 'mlp.fc2.bias'))
    x = F.layer_norm(x, (D,), g('norm.weight'), g('norm.bias'), eps)
    return x[:, 0], x[:, 1 + R:]


def resize_bicubic_antialias(x, size, antialias=True):
    """Restatement of kornia.geometry.transform.resize(..., interpolation='bicubic', align_corners=True, antialias=...) as published
    (kornia 0.6 / 0.7, geometry/transform/affwarp.py): when a side shrinks, blur first with a separable Gaussian of
    sigma = max((factor - 1) / 2, 0.001), kernel size max(int(4 sigma), 3) made odd, reflect border, then F.interpolate bicubic with
    align_corners=True.  kornia is a third-party dependency absent from this image: PARITY UNPINNED.  The HIP kernel ln3d_image_preprocess is
    tested against this function."""
    H, W = x.shape[-2:]
    fy, fx = H / size[0], W / size[1]
    if antialias and max(fy, fx) > 1:
        sig = (max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001))
        ks = [int(max(2.0 * 2 * s, 3)) for s in sig]
        ks = [k + 1 if k % 2 == 0 else k for k in ks]
        def g1(k, s):
            t = torch.arange(k, device=x.device, dtype=x.dtype) - k // 2
            w = torch.exp(-t * t / (2 * s * s))
            return w / w.sum()
        C = x.shape[1]
        ky, kx = g1(ks[0], sig[0]), g1(ks[1], sig[1])
        xp = F.pad(x, (ks[1] // 2, ks[1] // 2, ks[0] // 2, ks[0] // 2), mode='reflect')
        xp = F.conv2d(xp, kx.view(1, 1, 1, -1).expand(C, 1, 1, -1), groups=C)
        x = F.conv2d(xp, ky.view(1, 1, -1, 1).expand(C, 1, -1, 1), groups=C)
    return F.interpolate(x, size=size, mode='bicubic', align_corners=True)


def preprocess(x, S, mean, std, antialias=True):
    """the embedders' preprocess() (sgm/modules/encoders/modules.py:633-645,802-814)"""
    if tuple(x.shape[-2:]) != (S, S):
        x = resize_bicubic_antialias(x, (S, S), antialias)
    x = (x + 1.0) / 2.0
    return (x - torch.tensor(mean).to(x)[None, :, None, None]) / torch.tensor(std).to(x)[None, :, None, None]


def plucker_rays(c, S=224):
    """Pluecker ray maps of posed views: restatement of FrozenDinov2ImageEmbedderMVPlucker.gen_rays / get_plucker_ray
    (/root/reference/sgm/modules/encoders/modules.py:958-1005).  c [V, 25] = c2w 4x4 (row-major) + normalised intrinsics 3x3 ->
    [V, 6, S, S] = (o x d, d) per pixel centre.  Pinned against the reference's own methods by tests/golden/make_golden_mv.py."""
    out = []
    for cv in c.float():
        c2w, K = cv[:16].reshape(4, 4), cv[16:]
        yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float32) + 0.5, torch.arange(S, dtype=torch.float32) + 0.5, indexing='ij')
        xx, yy = (xx / S - K[2]) / K[0], (yy / S - K[5]) / K[4]
        d = torch.stack((xx, yy, torch.ones_like(xx)), -1)
        d = d / d.norm(dim=-1, keepdim=True)
        d = (c2w[None, :3, :3] @ d.reshape(-1, 3, 1))[..., 0].view(S, S, 3)
        o = c2w[None, :3, 3].expand(S * S, -1).reshape(S, S, 3)
        out.append(torch.cat([torch.cross(o, d, dim=-1), d], -1).permute(2, 0, 1))
    return torch.stack(out, 0)


def dinov2_mv_plucker_forward(sd, img_c, heads, n_cond_frames=4, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), size=224):
    """FrozenDinov2ImageEmbedderMVPlucker.forward (:1084-1111) for images already at the tower's size: the first n_cond_frames views,
    (x + 1) / 2 and mean / std on RGB, the 6 Pluecker maps appended UNNORMALISED (9 input channels), x_norm_patchtokens back as
    [B, T, L, D].  sd: dinov2 state dict with a 9-channel patch_embed.proj.weight."""
    img, c = img_c['img'].float(), img_c['c'].float()
    B, T = img.shape[0], n_cond_frames
    x = img[:, :T].reshape(B * T, *img.shape[2:])
    assert x.shape[-1] == size and x.shape[-2] == size, "resize first (resize_bicubic_antialias)"
    x = ((x + 1.0) / 2.0 - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    x = torch.cat([x, plucker_rays(c[:, :T].reshape(B * T, 25), size)], 1)
    _, tokens = dinov2_forward(sd, x, heads)
    return tokens.reshape(B, T, tokens.shape[1], tokens.shape[2])
