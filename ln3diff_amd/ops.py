"""Thin torch-tensor wrappers over the C ABI (include/ln3d.h).  torch is used for device
memory and streams only; every op below is a HIP kernel launch on torch's current stream."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import (EPI_F32, EPI_BF16, EPI_GELU_ERF, EPI_GELU_TANH, EPI_SILU, EPI_GATE_RES,  # noqa: F401
                   EPI_HEADS, EPI_F32_SILU, EPI_QUICK_GELU, EPI_CROSS_ATTN)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ln3diff_amd ops need device tensors (no CPU fallback exists)")


def reload_env():
    """Re-read the library's one measurement switch (LN3D_GEMM_TILE, the tile override the tests sweep: parsed once per process)."""
    L.lib().ln3d_reload_env()


def gemm(x, w, bias, epilogue, out0, out1=None, out2=None, *, M=None, ldo=None, gate=None, gate_rows=1,
         gate_ld=0, tokens=0, tok_pad=0, heads=0, head_dim=0, transpose_mask=0, head_dim_pad=0, ctx_keys=0, ctx_pad=0,
         ctx_scale=0.0, head_norm0=None, head_norm1=None, head_norm_eps=1e-5, res_bias=None, res_bias_ld=0):
    """out = epi(x[M,K] @ w[N,K]^T + bias).  x, w bf16 (row stride = shape[-1])."""
    _chk_dev(x, w, out0)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    a = L.GemmArgs()
    K = w.shape[1]
    if x.shape[-1] != K:
        raise ValueError(f"gemm: activation has {x.shape[-1]} input features, the weight expects {K}")
    a.X, a.ldx, a.W, a.ldw = _p(x), x.stride(-2) if x.dim() > 1 else K, _p(w), w.stride(0)
    a.bias = _p(bias)
    a.M = int(M if M is not None else x.numel() // K)
    a.N, a.K = w.shape[0], K
    a.epilogue = epilogue
    a.out0, a.out1, a.out2 = _p(out0), _p(out1), _p(out2)
    a.ldo = int(ldo if ldo is not None else w.shape[0])
    a.gate, a.gate_rows, a.gate_ld = _p(gate), gate_rows, gate_ld
    a.tokens, a.tok_pad, a.heads, a.head_dim, a.transpose_mask = tokens, tok_pad, heads, head_dim, transpose_mask
    a.head_dim_pad = head_dim_pad
    a.ctx_keys, a.ctx_pad, a.ctx_scale = ctx_keys, ctx_pad, float(ctx_scale)
    a.head_norm0, a.head_norm1, a.head_norm_eps = _p(head_norm0), _p(head_norm1), float(head_norm_eps)
    a.res_bias, a.res_bias_ld = _p(res_bias), int(res_bias_ld)
    L.check(L.lib().ln3d_gemm_bf16(C.byref(a), _stream()), "gemm")


def heads_norm_fusable(M, N, tokens, head_dim, head_dim_pad=0):
    """True when ln3d_gemm_bf16's HEADS epilogue applies qk_norm itself for this problem - the library's own answer (it depends on
    the tile configuration it picks), not a copy of its heuristic."""
    return bool(L.lib().ln3d_gemm_heads_norm_fusable(int(M), int(N), int(tokens), int(head_dim), int(head_dim_pad)))


def attention(q, k, vt, out, B, H, Nq, Nq_pad, Nk, Nk_pad, Dh, scale=None, causal=False, dh_true=0):
    """dh_true: true head size when q / k / vt rows are zero-padded to Dh; the output is then compact [B, Nq, H * dh_true]."""
    _chk_dev(q, k, vt, out)
    a = L.AttnArgs()
    a.Q, a.K, a.Vt, a.O = _p(q), _p(k), _p(vt), _p(out)
    a.B, a.H, a.Nq, a.Nq_pad, a.Nk, a.Nk_pad, a.Dh = B, H, Nq, Nq_pad, Nk, Nk_pad, Dh
    a.Dh_true = int(dh_true)
    a.ldo = H * (dh_true if dh_true and dh_true != Dh else Dh)
    a.scale = float(scale if scale is not None else Dh ** -0.5)
    a.causal = int(bool(causal))
    L.check(L.lib().ln3d_attention_bf16(C.byref(a), _stream()), "attention")


def rmsnorm_heads(x, w, rows, Dh, eps=1e-5, true_dim=0):
    """x rows of Dh (64 / 128) bf16 in place; true_dim < Dh when heads are zero-padded (w padded with zeros to Dh)."""
    L.check(L.lib().ln3d_rmsnorm_heads_bf16(_p(x), _p(w), C.c_int64(rows), Dh, int(true_dim), C.c_float(eps), _stream()), "rmsnorm_heads")


def norm_modulate(x, y, rows, D, kind=0, eps=1e-6, weight=None, shift=None, scale=None, mod_rows=1, mod_ld=0,
                  shift_table=None, scale_table=None, rows_in=0, rows_out=0):
    _chk_dev(x, y)
    a = L.NormArgs()
    a.x, a.y, a.rows, a.D, a.kind, a.eps, a.weight = _p(x), _p(y), rows, D, kind, eps, _p(weight)
    a.shift, a.scale, a.mod_rows, a.mod_ld = _p(shift), _p(scale), mod_rows, mod_ld
    a.shift_table, a.scale_table, a.rows_in, a.rows_out = _p(shift_table), _p(scale_table), rows_in, rows_out
    L.check(L.lib().ln3d_norm_modulate(C.byref(a), _stream()), "norm_modulate")


def timestep_embedding(t, out, B, dim=256):
    L.check(L.lib().ln3d_timestep_embedding(_p(t), _p(out), B, dim, _stream()), "timestep_embedding")


def add_act_cast(a, b, y_bf16, sum_f32, n, act):
    L.check(L.lib().ln3d_add_act_cast(_p(a), _p(b), _p(y_bf16), _p(sum_f32), C.c_int64(n), act, _stream()), "add_act_cast")


def cast_bf16(x, y):
    L.check(L.lib().ln3d_cast_f32_bf16(_p(x), _p(y), C.c_int64(x.numel()), _stream()), "cast")


def patch_embed(x, in_scale, w, bias, pos, tokens, Bx, Bn, Cc, S, p, D):
    L.check(L.lib().ln3d_patch_embed(_p(x), _p(in_scale), _p(w), _p(bias), _p(pos), _p(tokens), Bx, Bn, Cc, S, p, D, _stream()),
            "patch_embed")


def final_layer(tokens, shift, scale, mod_ld, shift_table, scale_table, w, bias, out, Bn, Cc, S, p, D):
    L.check(L.lib().ln3d_final_layer(_p(tokens), _p(shift), _p(scale), C.c_int64(mod_ld), _p(shift_table), _p(scale_table),
                                     _p(w), _p(bias), _p(out), Bn, Cc, S, p, D, _stream()), "final_layer")


def edm_euler_step(x, eps2, sigma, sigma_next, cfg_scale):
    L.check(L.lib().ln3d_edm_euler_step(_p(x), _p(eps2), C.c_float(sigma), C.c_float(sigma_next), C.c_float(cfg_scale),
                                        C.c_int64(x.numel()), _stream()), "edm_euler_step")


def ddpm_step(x, eps, noise, a, b, c1, c2, sig, clip):
    L.check(L.lib().ln3d_ddpm_step(_p(x), _p(eps), _p(noise), C.c_float(a), C.c_float(b), C.c_float(c1), C.c_float(c2),
                                   C.c_float(sig), int(clip), C.c_int64(x.numel()), _stream()), "ddpm_step")


def flow_euler_step(x2, v2, dt, cfg_scale):
    L.check(L.lib().ln3d_flow_euler_step(_p(x2), _p(v2), C.c_float(dt), C.c_float(cfg_scale), C.c_int64(x2.numel() // 2),
                                         _stream()), "flow_euler_step")


def axpby(x, y, a, b):
    L.check(L.lib().ln3d_axpby(_p(x), _p(y), C.c_float(a), C.c_float(b), C.c_int64(x.numel()), _stream()), "axpby")


def planes_to_channel_last(src, dst, NP, Cc, H, W):
    L.check(L.lib().ln3d_planes_to_channel_last(_p(src), _p(dst), NP, Cc, H, W, _stream()), "planes_to_channel_last")


def planes_to_nchw(src, dst, NP, Cc, H, W):
    L.check(L.lib().ln3d_planes_to_nchw(_p(src), _p(dst), NP, Cc, H, W, _stream()), "planes_to_nchw")


def render_triplane(planes_cl, H, W, plane_index, cams, res, dec, jitter, u_fine, rgb, depth, wsum, ray_limits, scalars,
                    box_warp=0.9, bbox_min=-0.45, bbox_max=0.45, white_back=True, coarse_sigma=None, fine_depths=None,
                    ray_o=None, ray_d=None, fine_sigma=None, coarse_coords=None, fine_coords=None, n_views=None, views_per_call=0,
                    rays_per_view=0, visibility=None, depth_resolution=0, depth_resolution_importance=0, ray_start='auto', ray_end='auto',
                    filter_out_of_bbox=True, weights=None, all_coords=None, feature_volume=None):
    """cams [V,25] (rays generated in-kernel) or explicit ray_o / ray_d [V, M, 3] (then cams may be None; rays_per_view = M).
    views_per_call: how many consecutive views form one reference forward() call for the call-wide reductions (ray-limit fix-up,
    depth clamp range); 0 = all of them (include/ln3d.h).  ray_start / ray_end: both 'auto' or both numbers."""
    a = L.RenderArgs()
    a.planes, a.H, a.W, a.plane_index, a.cams = _p(planes_cl), H, W, _p(plane_index), _p(cams)
    a.V, a.res = (cams.shape[0] if cams is not None else n_views), res
    a.dec_w0, a.dec_b0, a.dec_w1, a.dec_b1 = (_p(t) for t in dec)
    a.jitter, a.u_fine = _p(jitter), _p(u_fine)
    a.box_warp, a.bbox_min, a.bbox_max, a.white_back = box_warp, bbox_min, bbox_max, int(white_back)
    a.rgb, a.depth, a.wsum, a.ray_limits, a.scalars = _p(rgb), _p(depth), _p(wsum), _p(ray_limits), _p(scalars)
    a.coarse_sigma, a.fine_depths = _p(coarse_sigma), _p(fine_depths)
    a.ray_o, a.ray_d, a.fine_sigma, a.coarse_coords, a.fine_coords = _p(ray_o), _p(ray_d), _p(fine_sigma), _p(coarse_coords), _p(fine_coords)
    a.views_per_call = int(views_per_call)
    a.rays_per_view, a.visibility = int(rays_per_view), _p(visibility)
    a.depth_resolution, a.depth_resolution_importance = int(depth_resolution), int(depth_resolution_importance)
    if (ray_start == 'auto') != (ray_end == 'auto'):
        raise ValueError("ray_start / ray_end: both 'auto' or both numbers (renderer.py:145)")
    if ray_start == 'auto':
        a.ray_mode = 0
    else:
        a.ray_mode, a.ray_start, a.ray_end = 1, float(ray_start), float(ray_end)
    a.no_bbox_filter = 0 if filter_out_of_bbox else 1
    a.weights, a.all_coords, a.feature_volume = _p(weights), _p(all_coords), _p(feature_volume)
    L.check(L.lib().ln3d_render_triplane(C.byref(a), _stream()), "render_triplane")


def query_points(planes_cl, H, W, points, dec, box_warp, sigma, rgb, scalars):
    """scalars: caller-owned f32 scratch of _lib.RENDER_SCRATCH_FLOATS (no allocation inside the library)."""
    L.check(L.lib().ln3d_query_points(_p(planes_cl), H, W, _p(points), C.c_int64(points.shape[0]), *(_p(t) for t in dec),
                                      C.c_float(box_warp), _p(sigma), _p(rgb), _p(scalars), _stream()), "query_points")


def groupnorm_swish(x, w, b, y, stats, N, HW, Cc, groups=32, eps=1e-6, swish=True):
    L.check(L.lib().ln3d_groupnorm_swish(_p(x), _p(w), _p(b), _p(y), _p(stats), N, HW, Cc, groups, C.c_float(eps), int(swish),
                                         _stream()), "groupnorm_swish")


def im2col3x3(x, col, N, H, W, Cc, upsample, Kpad):
    L.check(L.lib().ln3d_im2col3x3(_p(x), _p(col), N, H, W, Cc, upsample, Kpad, _stream()), "im2col3x3")


def patch_embed_triplane(latent, w, bias, out_silu, out_raw, B, Cg, S, p, D):
    L.check(L.lib().ln3d_patch_embed_triplane(_p(latent), _p(w), _p(bias), _p(out_silu), _p(out_raw), B, Cg, S, p, D, _stream()),
            "patch_embed_triplane")


def tile_rows(x, y, per, reps):
    L.check(L.lib().ln3d_tile_rows(_p(x), _p(y), C.c_int64(per), reps, _stream()), "tile_rows")


def add_table_rows(t0, tables, out, layers, B, W):
    L.check(L.lib().ln3d_add_table_rows(_p(t0), _p(tables), _p(out), layers, B, C.c_int64(W), _stream()), "add_table_rows")


def cfg_combine_dup(v2, cfg_scale):
    L.check(L.lib().ln3d_cfg_combine_dup(_p(v2), C.c_float(cfg_scale), C.c_int64(v2.numel() // 2), _stream()), "cfg_combine_dup")


def vt_key_order(n_pad, device=None):
    """Index map of the V^T layout consumed by ln3d_attention_bf16: position p of every 16-key group holds key
    perm(p) with bits 2 and 3 swapped ([0-3, 8-11, 4-7, 12-15]).  ln3d_gemm_bf16's HEADS epilogue writes this layout
    itself; this helper exists for callers (tests, op-level bindings) that build V^T by hand: vt_perm = vt[..., idx]."""
    t = torch.arange(n_pad, device=device)
    return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1)


def ddim_step(x, eps_u, eps_c, noise, cfg_scale, a, b, sqrt_ab_prev, coef_eps, sigma, clip):
    L.check(L.lib().ln3d_ddim_step(_p(x), _p(eps_u), _p(eps_c), _p(noise), C.c_float(cfg_scale), C.c_float(a), C.c_float(b),
                                   C.c_float(sqrt_ab_prev), C.c_float(coef_eps), C.c_float(sigma), int(clip),
                                   C.c_int64(x.numel()), _stream()), "ddim_step")


def mesh_count(sigma, G, thr, counts):
    L.check(L.lib().ln3d_mesh_count(_p(sigma), G, C.c_float(thr), _p(counts), _stream()), "mesh_count")


def mesh_emit(sigma, G, thr, offsets, tri_pos, tri_key):
    L.check(L.lib().ln3d_mesh_emit(_p(sigma), G, C.c_float(thr), _p(offsets), _p(tri_pos), _p(tri_key), _stream()), "mesh_emit")


def mcubes_count(sigma, G, thr, counts):
    L.check(L.lib().ln3d_mcubes_count(_p(sigma), G, C.c_float(thr), _p(counts), _stream()), "mcubes_count")


def mcubes_emit(sigma, G, thr, offsets, tri_pos, tri_key):
    L.check(L.lib().ln3d_mcubes_emit(_p(sigma), G, C.c_float(thr), _p(offsets), _p(tri_pos), _p(tri_key), _stream()), "mcubes_emit")


def lincomb(y, ks, cs, out):
    n = len(ks)
    arr_k = (C.c_void_p * n)(*[k.data_ptr() for k in ks])
    arr_c = (C.c_float * n)(*[float(c) for c in cs])
    L.check(L.lib().ln3d_lincomb(_p(y), arr_k, arr_c, n, _p(out), C.c_int64(out.numel()), _stream()), "lincomb")


def err_ratio_sq(err, y0, y1, atol, rtol, acc):
    L.check(L.lib().ln3d_err_ratio_sq(_p(err), _p(y0), _p(y1), C.c_float(atol), C.c_float(rtol), _p(acc),
                                      C.c_int64(err.numel()), _stream()), "err_ratio_sq")


def embed_tokens(ids, tok_emb, pos_emb, out, B, T, D):
    assert ids.dtype == torch.int32
    L.check(L.lib().ln3d_embed_tokens(_p(ids), _p(tok_emb), _p(pos_emb), _p(out), B, T, D, tok_emb.shape[0], _stream()), "embed_tokens")


def layernorm_f32(x, w, b, y, rows, D, eps=1e-5):
    L.check(L.lib().ln3d_layernorm_f32(_p(x), _p(w), _p(b), _p(y), C.c_int64(rows), D, C.c_float(eps), _stream()), "layernorm_f32")


def image_preprocess(x, S, antialias, mean, std):
    """[N, C, H, W] f32 in [-1, 1] -> resized (kornia bicubic, align_corners, optional antialias blur) and normalised [N, C, S, S]"""
    _chk_dev(x)
    N, Cc, H, W = x.shape
    x = x.contiguous().float()
    out = torch.empty(N, Cc, S, S, device=x.device, dtype=torch.float32)
    tmp = torch.empty(2 * x.numel(), device=x.device, dtype=torch.float32) if (antialias and (H > S or W > S)) else None
    m = (C.c_float * Cc)(*[float(v) for v in mean])
    sd = (C.c_float * Cc)(*[float(v) for v in std])
    L.check(L.lib().ln3d_image_preprocess(_p(x), _p(out), _p(tmp), N, Cc, H, W, S, int(bool(antialias)), m, sd, _stream()), "image_preprocess")
    return out


def vit_patchify(img, out, B, S, p, Kpad, C=3):
    L.check(L.lib().ln3d_vit_patchify(_p(img), _p(out), B, S, p, Kpad, C, _stream()), "vit_patchify")


def plucker_rays(c, S):
    """c f32 [V, 25] (c2w 4x4 + intrinsics 3x3) -> Pluecker maps f32 [V, 6, S, S] (o x d, d)."""
    _chk_dev(c)
    V = c.shape[0]
    out = torch.empty(V, 6, S, S, device=c.device, dtype=torch.float32)
    c32 = c.contiguous().float()                     # kept alive across the launch (a temporary's block could be handed out again)
    L.check(L.lib().ln3d_plucker_rays(_p(c32), _p(out), V, S, _stream()), "plucker_rays")
    return out


def vit_assemble(patch, cls, reg, pos, x, B, Lp, R, D):
    L.check(L.lib().ln3d_vit_assemble(_p(patch), _p(cls), _p(reg), _p(pos), _p(x), B, Lp, R, D, _stream()), "vit_assemble")


def device_cus():
    return int(L.lib().ln3d_device_cus())


def probe_mfma(out, wgs, iters):
    """Diagnostic pure-MFMA stream (include/ln3d.h ln3d_probe_mfma_bf16); returns the flop it executes."""
    assert out.is_cuda and out.dtype == torch.float32 and out.numel() >= wgs * 512
    L.check(L.lib().ln3d_probe_mfma_bf16(_p(out), int(wgs), int(iters), _stream()), "probe_mfma")
    return wgs * 8 * iters * 8 * 2.0 * 32 * 32 * 16


# ---------------------------------------------------------------- U-Net pieces (csrc/unet_ops.hip)
def groupnorm_any(x, w, b, y, N, HW, Cc, groups=32, eps=1e-5, swish=True, add_row=None, mod_scale=None, mod_shift=None):
    _chk_dev(x, y)
    L.check(L.lib().ln3d_groupnorm_any(_p(x), _p(add_row), _p(w), _p(b), _p(mod_scale), _p(mod_shift), _p(y), N, HW, Cc, groups, C.c_float(eps),
                                       int(swish), _stream()), "groupnorm_any")


def im2col3x3_strided(x, col, N, H, W, Cc, stride, Kpad):
    L.check(L.lib().ln3d_im2col3x3_strided(_p(x), _p(col), N, H, W, Cc, stride, Kpad, _stream()), "im2col3x3_strided")


def geglu(x, y, rows, inner):
    L.check(L.lib().ln3d_geglu(_p(x), _p(y), C.c_int64(rows), inner, _stream()), "geglu")


def attention_small(q, k, v, out, B, H, Nq, Nk, Dh, ldq, ldk, ldv, scale):
    """q / k / v: bf16 tensors (or views into wider projection outputs) whose data_ptr is column 0 of head 0; ld* = their row strides"""
    _chk_dev(q, k, v, out)
    L.check(L.lib().ln3d_attention_small(_p(q), _p(k), _p(v), _p(out), B, H, Nq, Nk, Dh, C.c_int64(ldq), C.c_int64(ldk), C.c_int64(ldv),
                                         C.c_float(scale), _stream()), "attention_small")


def nchw_to_cl_bf16(x, y, N, Cc, HW, Cpad):
    L.check(L.lib().ln3d_nchw_to_cl_bf16(_p(x), _p(y), N, Cc, HW, Cpad, _stream()), "nchw_to_cl_bf16")


def cl_to_nchw_f32(x, y, N, Cc, HW):
    L.check(L.lib().ln3d_cl_to_nchw_f32(_p(x), _p(y), N, Cc, HW, _stream()), "cl_to_nchw_f32")


def mix_prediction(eps, x, mixing_logit, sqrt_one_minus_ab, N, Cc, HW):
    L.check(L.lib().ln3d_mix_prediction(_p(eps), _p(x), _p(mixing_logit), C.c_float(sqrt_one_minus_ab), N, Cc, HW, _stream()), "mix_prediction")
