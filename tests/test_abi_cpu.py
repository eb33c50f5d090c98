"""C-ABI contract without a GPU: every entry point of include/ln3d.h rejects missing buffers with LN3D_ERR_BAD_ARG BEFORE it touches
the device (argument validation is not a compute call; nothing is launched here).  The table must cover every `int ln3d_*(...)`
declared in the header, so a new export without validation shows up as a failing test."""
import ctypes as C
import os
import re

from conftest import ROOT

N = None
F, I64 = C.c_float, C.c_int64
NULL_CALLS = {
    'ln3d_gemm_bf16': (N, N),
    'ln3d_attention_bf16': (N, N),
    'ln3d_norm_modulate': (N, N),
    'ln3d_render_triplane': (N, N),
    'ln3d_embed_tokens': (N, N, N, N, 1, 1, 4, 1, N),
    'ln3d_layernorm_f32': (N, N, N, N, I64(1), 128, F(1e-5), N),
    'ln3d_vit_patchify': (N, N, 1, 224, 14, 640, 3, N),
    'ln3d_plucker_rays': (N, N, 1, 224, N),
    'ln3d_vit_assemble': (N, N, N, N, N, 1, 1, 0, 128, N),
    'ln3d_image_preprocess': (N, N, N, 1, 3, 8, 8, 4, 1, N, N, N),
    'ln3d_rmsnorm_heads_bf16': (N, N, I64(1), 64, 64, F(1e-5), N),
    'ln3d_timestep_embedding': (N, N, 1, 256, N),
    'ln3d_add_act_cast': (N, N, N, N, I64(1), 0, N),
    'ln3d_cast_f32_bf16': (N, N, I64(1), N),
    'ln3d_patch_embed': (N, N, N, N, N, N, 1, 1, 4, 32, 2, 128, N),
    'ln3d_patch_embed_triplane': (N, N, N, N, N, 1, 4, 32, 2, 128, N),
    'ln3d_tile_rows': (N, N, I64(1), 1, N),
    'ln3d_final_layer': (N, N, N, I64(0), N, N, N, N, N, 1, 4, 32, 2, 128, N),
    'ln3d_edm_euler_step': (N, N, F(1), F(1), F(1), I64(1), N),
    'ln3d_ddpm_step': (N, N, N, F(1), F(1), F(1), F(1), F(1), 0, I64(1), N),
    'ln3d_ddim_step': (N, N, N, N, F(1), F(1), F(1), F(1), F(1), F(1), 0, I64(1), N),
    'ln3d_flow_euler_step': (N, N, F(1), F(1), I64(1), N),
    'ln3d_add_table_rows': (N, N, N, 1, 1, I64(4), N),
    'ln3d_cfg_combine_dup': (N, F(1), I64(1), N),
    'ln3d_lincomb': (N, N, N, 1, N, I64(1), N),
    'ln3d_err_ratio_sq': (N, N, N, F(1), F(1), N, I64(1), N),
    'ln3d_axpby': (N, N, F(1), F(1), I64(1), N),
    'ln3d_planes_to_channel_last': (N, N, 1, 32, 8, 8, N),
    'ln3d_planes_to_nchw': (N, N, 1, 32, 8, 8, N),
    'ln3d_query_points': (N, 8, 8, N, I64(1), N, N, N, N, F(0.9), N, N, N, N),
    'ln3d_mesh_count': (N, 8, F(1), N, N),
    'ln3d_mcubes_count': (N, 8, F(1), N, N),
    'ln3d_mcubes_emit': (N, 8, F(1), N, N, N, N),
    'ln3d_mesh_emit': (N, 8, F(1), N, N, N, N),
    'ln3d_groupnorm_swish': (N, N, N, N, N, 1, 64, 64, 32, F(1e-6), 1, N),
    'ln3d_im2col3x3': (N, N, 1, 8, 8, 64, 1, 576, N),
    'ln3d_probe_mfma_bf16': (N, 1, 1, N),
    'ln3d_groupnorm_any': (N, N, N, N, N, N, N, 1, 16, 64, 32, F(1e-5), 1, N),
    'ln3d_im2col3x3_strided': (N, N, 1, 8, 8, 64, 2, 576, N),
    'ln3d_geglu': (N, N, I64(1), 64, N),
    'ln3d_attention_small': (N, N, N, N, 1, 1, 16, 16, 40, I64(40), I64(40), I64(40), F(0.1), N),
    'ln3d_nchw_to_cl_bf16': (N, N, 1, 12, 64, 16, N),
    'ln3d_cl_to_nchw_f32': (N, N, 1, 12, 64, N),
    'ln3d_mix_prediction': (N, N, N, F(0.5), 1, 12, 64, N),
}
NOT_A_KERNEL = {'ln3d_abi_version', 'ln3d_gemm_heads_norm_fusable', 'ln3d_device_cus'}      # pure host queries


def test_every_entry_point_rejects_missing_buffers(hip_lib):
    hdr = open(os.path.join(ROOT, 'include', 'ln3d.h')).read()
    declared = set(re.findall(r'^int (ln3d_[a-z0-9_]+)\(', hdr, re.M))
    assert declared - NOT_A_KERNEL == set(NULL_CALLS), (declared - NOT_A_KERNEL) ^ set(NULL_CALLS)
    for name, args in NULL_CALLS.items():
        rc = getattr(hip_lib, name)(*args)
        assert rc == -1, (name, rc)                                       # LN3D_ERR_BAD_ARG
    assert b'bad argument' in hip_lib.ln3d_strerror(-1)


def test_host_queries(hip_lib):
    assert hip_lib.ln3d_abi_version() == 10
    # the fused qk-norm epilogue needs head-aligned tiles: 64-wide heads yes, 72-in-128 padded heads no
    assert hip_lib.ln3d_gemm_heads_norm_fusable(12288, 3072, 768, 64, 64) == 1
    assert hip_lib.ln3d_gemm_heads_norm_fusable(12288, 3 * 16 * 128, 768, 72, 128) == 0


def test_gate_residual_row_arguments_are_validated(hip_lib):
    """the GATE_RES epilogue reads float4 quads at gate / res_bias + (row / gate_rows) * ld + feature - a leading dimension
    that is not a multiple of 4, a misaligned pointer or gate_rows left at 0 next to a per-sample buffer is a bad argument, not an
    out-of-bounds read.  Validation happens before anything touches the device (fake, never dereferenced addresses)."""
    from ln3diff_amd._lib import GemmArgs, EPI_GATE_RES
    base = dict(X=0x10000, ldx=64, W=0x20000, ldw=64, M=192, N=256, K=64, epilogue=EPI_GATE_RES, out0=0x30000, ldo=256)
    bad = [dict(gate=0x40000, gate_rows=96, gate_ld=258),              # ld % 4
           dict(gate=0x40004, gate_rows=96, gate_ld=256),              # pointer not 16-byte aligned
           dict(gate=0x40000, gate_rows=0, gate_ld=256),               # rows per sample missing
           dict(res_bias=0x50000, res_bias_ld=255, gate_rows=96),
           dict(res_bias=0x50008, res_bias_ld=256, gate_rows=96),
           dict(res_bias=0x50000, res_bias_ld=256, gate_rows=0)]
    for extra in bad:
        a = GemmArgs(**dict(base, **extra))
        assert hip_lib.ln3d_gemm_bf16(C.byref(a), None) == -1, extra
