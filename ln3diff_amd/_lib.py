"""ctypes binding of libln3d_hip.so (the C ABI of include/ln3d.h).

There is NO fallback: if the library is missing or a kernel launch fails the product raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libln3d_hip.so")

SYMBOLS = [
    "ln3d_strerror", "ln3d_abi_version", "ln3d_reload_env", "ln3d_gemm_bf16", "ln3d_gemm_heads_norm_fusable", "ln3d_attention_bf16",
    "ln3d_rmsnorm_heads_bf16", "ln3d_norm_modulate", "ln3d_timestep_embedding",
    "ln3d_add_act_cast", "ln3d_cast_f32_bf16", "ln3d_patch_embed", "ln3d_final_layer",
    "ln3d_edm_euler_step", "ln3d_ddpm_step", "ln3d_flow_euler_step", "ln3d_axpby",
    "ln3d_planes_to_channel_last", "ln3d_planes_to_nchw", "ln3d_render_triplane",
    "ln3d_query_points", "ln3d_groupnorm_swish", "ln3d_im2col3x3", "ln3d_patch_embed_triplane", "ln3d_tile_rows", "ln3d_add_table_rows", "ln3d_cfg_combine_dup", "ln3d_ddim_step", "ln3d_mesh_count", "ln3d_mesh_emit", "ln3d_mcubes_count", "ln3d_mcubes_emit", "ln3d_lincomb", "ln3d_err_ratio_sq", "ln3d_embed_tokens", "ln3d_layernorm_f32", "ln3d_vit_patchify", "ln3d_vit_assemble", "ln3d_image_preprocess", "ln3d_plucker_rays",
    "ln3d_device_cus", "ln3d_probe_mfma_bf16",
    "ln3d_groupnorm_any", "ln3d_im2col3x3_strided", "ln3d_geglu", "ln3d_attention_small", "ln3d_nchw_to_cl_bf16", "ln3d_cl_to_nchw_f32",
    "ln3d_mix_prediction",
]

EPI_F32, EPI_BF16, EPI_GELU_ERF, EPI_GELU_TANH, EPI_SILU, EPI_GATE_RES, EPI_HEADS, EPI_F32_SILU, EPI_QUICK_GELU, EPI_CROSS_ATTN = range(10)
RENDER_SCRATCH_FLOATS = 16384
RENDER_MAX_CALLS = (RENDER_SCRATCH_FLOATS - 2644) // 8      # per-call range records after the decoder image (csrc/render.hip)

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float


class GemmArgs(C.Structure):
    _fields_ = [("X", vp), ("ldx", i64), ("W", vp), ("ldw", i64), ("bias", vp),
                ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32),
                ("out0", vp), ("out1", vp), ("out2", vp), ("ldo", i64),
                ("gate", vp), ("gate_rows", i32), ("gate_ld", i64),
                ("tokens", i32), ("tok_pad", i32), ("heads", i32), ("head_dim", i32),
                ("transpose_mask", i32), ("ctx_keys", i32), ("ctx_pad", i32), ("ctx_scale", f32), ("head_dim_pad", i32),
                ("head_norm0", vp), ("head_norm1", vp), ("head_norm_eps", f32),
                ("res_bias", vp), ("res_bias_ld", i64)]


class AttnArgs(C.Structure):
    _fields_ = [("Q", vp), ("K", vp), ("Vt", vp), ("O", vp),
                ("B", i32), ("H", i32), ("Nq", i32), ("Nq_pad", i32), ("Nk", i32), ("Nk_pad", i32),
                ("Dh", i32), ("ldo", i64), ("scale", f32), ("causal", i32), ("Dh_true", i32)]


class NormArgs(C.Structure):
    _fields_ = [("x", vp), ("y", vp), ("rows", i64), ("D", i32), ("kind", i32), ("eps", f32),
                ("weight", vp), ("shift", vp), ("scale", vp), ("mod_rows", i32), ("mod_ld", i64),
                ("shift_table", vp), ("scale_table", vp), ("rows_in", i32), ("rows_out", i32)]


class RenderArgs(C.Structure):
    _fields_ = [("planes", vp), ("H", i32), ("W", i32), ("plane_index", vp), ("cams", vp),
                ("V", i32), ("res", i32), ("dec_w0", vp), ("dec_b0", vp), ("dec_w1", vp), ("dec_b1", vp),
                ("jitter", vp), ("u_fine", vp), ("box_warp", f32), ("bbox_min", f32), ("bbox_max", f32),
                ("white_back", i32), ("rgb", vp), ("depth", vp), ("wsum", vp), ("ray_limits", vp),
                ("scalars", vp), ("coarse_sigma", vp), ("fine_depths", vp), ("ray_o", vp), ("ray_d", vp),
                ("fine_sigma", vp), ("coarse_coords", vp), ("fine_coords", vp), ("views_per_call", i32),
                ("rays_per_view", i32), ("visibility", vp), ("depth_resolution", i32), ("depth_resolution_importance", i32),
                ("ray_mode", i32), ("ray_start", f32), ("ray_end", f32), ("no_bbox_filter", i32),
                ("weights", vp), ("all_coords", vp), ("feature_volume", vp)]


_lib = None


def lib():
    """dlopen the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python __graft_entry__.py`).  ln3diff_amd has no CPU/eager fallback.")
        # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  It must be in the process BEFORE this library is
        # loaded, so that our NEEDED libamdhip64.so.7 resolves to the same runtime; loaded the other way round (build() then
        # smoke() in one process) /opt/rocm's copy comes in first, torch then loads its own, and launches on torch's streams
        # fail with "HIP kernel launch failed".
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        _lib.ln3d_strerror.restype = C.c_char_p
        for s in SYMBOLS:
            if s != "ln3d_strerror":
                getattr(_lib, s).restype = C.c_int
    return _lib


def check_symbols():
    L = lib()
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError(f"libln3d_hip.so lacks symbols: {missing}")
    assert L.ln3d_abi_version() == 10
    return True


def check(code, what=""):
    if code != 0:
        raise RuntimeError(f"ln3d {what} failed: {lib().ln3d_strerror(code).decode()} ({code})")
