"""`ImportanceRenderer` - the lower seam of the reference's renderer (nsr/volumetric_rendering/renderer.py:125-307) on the fused
HIP ray-marcher: `forward(planes, decoder, ray_origins, ray_directions, rendering_options, return_meta)` with explicit rays.

Everything the reference does between this call and its return - AABB ray limits (math_utils.get_ray_limits_box), the invalid-ray
fix-up, stratified sampling, `sample_from_planes` x3, the decoder MLP, MipRayMarcher2, importance sampling, `unify_samples`
and the second ray-march - runs inside `ln3d_render_triplane` (csrc/render.hip) in its explicit-ray mode.  Random numbers are
inputs (`jitter` [N, M, 64] for `sample_stratified`, `u_fine` [N*M, 64] for `sample_pdf`); `draw_render_noise` restates the
reference's RNG consumption order.

Supported rendering_options: every sampling preset of nsr/script_util.py:433-1000 - depth_resolution / depth_resolution_importance
up to 128 each (presets use 48, 64, 80, 96, 128), ray_start = ray_end = 'auto' (ray / AABB limits) or two numbers (ShapeNet / FFHQ
presets), filter_out_of_bbox on or off, clamp_mode 'softplus', white_back on or off (default True like ray_marcher.py:63).  The
Objaverse preset (64 + 64, 'auto', bbox filter) runs the lane = sample kernel, everything else - and `return_meta`, whose merged
per-sample tensors 'weights' / 'all_coords' / 'feature_volume' then exist in memory - the generic kernel of csrc/render.hip.
'visibility' (T behind the last interval) is always returned.  M, the number of rays per batch element, is arbitrary.
Not built: disparity_space_sampling, density_noise, return_surface, depth_resolution_importance = 0 (no released preset uses them).
"""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import RENDER_SCRATCH_FLOATS


def draw_render_noise(V, M, S=64, generator=None, device='cpu', n_importance=None):
    """The reference's RNG consumption per Triplane.forward, as logical tensors:
    coarse jitter = rand_like on a [S,V,M,1]-strided tensor (the 'auto' branch; the numeric branch draws on a contiguous [V,M,S,1]),
    then fine uniforms rand(V*M, N_importance)."""
    NI = S if n_importance is None else n_importance
    if device == 'cpu' or str(device) == 'cpu':
        j = torch.rand(S, V, M, 1, generator=generator).permute(1, 2, 0, 3).reshape(V, M, S).contiguous()
        u = torch.rand(V * M, NI, generator=generator)
    else:
        j = torch.rand(V, M, S, device=device, generator=generator)
        u = torch.rand(V * M, NI, device=device, generator=generator)
    return j, u


MAX_SAMPLES = 128       # per pass (csrc/render.hip GEN_MAXS)


def check_rendering_options(rk):
    S, NI = rk.get('depth_resolution', 64), rk.get('depth_resolution_importance', 64)
    if not (4 <= S <= MAX_SAMPLES and 1 <= NI <= MAX_SAMPLES):
        raise NotImplementedError("the HIP renderer takes 4..%d coarse and 1..%d importance samples per ray (got %r + %r)" % (MAX_SAMPLES, MAX_SAMPLES, S, NI))
    rs, re_ = rk.get('ray_start'), rk.get('ray_end')
    if (rs == 'auto') != (re_ == 'auto') or (rs != 'auto' and not (float(re_) > float(rs))):
        raise ValueError("ray_start / ray_end: both 'auto' or two numbers with ray_end > ray_start (renderer.py:145-163)")
    if rk.get('clamp_mode', 'softplus') != 'softplus':
        raise NotImplementedError("clamp_mode='softplus' only (so does MipRayMarcher2, ray_marcher.py:36)")
    if rk.get('disparity_space_sampling', False) or rk.get('density_noise', 0) > 0 or rk.get('return_surface', False):
        raise NotImplementedError("disparity-space sampling / density noise / return_surface are not part of the sampling hot path")


def render_call_kwargs(rk):
    """The preset arguments of ops.render_triplane from a rendering_options dict."""
    fb = bool(rk.get('filter_out_of_bbox', False))
    return dict(box_warp=rk['box_warp'], bbox_min=rk.get('sampler_bbox_min', -0.5 * rk['box_warp']) if fb else 0.0,
                bbox_max=rk.get('sampler_bbox_max', 0.5 * rk['box_warp']) if fb else 0.0, white_back=rk.get('white_back', True),
                depth_resolution=rk.get('depth_resolution', 64), depth_resolution_importance=rk.get('depth_resolution_importance', 64),
                ray_start=rk['ray_start'], ray_end=rk['ray_end'], filter_out_of_bbox=fb)


def decoder_weights(decoder, device):
    """(w0, b0, w1, b1) of an OSGDecoder-shaped module (`net.0`, `net.2` FullyConnectedLayers: 32 -> 64 -> 1 + 3)."""
    n = decoder.net
    if tuple(n[0].weight.shape) != (64, 32) or n[2].weight.shape[1] != 64 or n[2].weight.shape[0] < 4:
        raise NotImplementedError("the HIP renderer is built for the released 32 -> 64 -> (1 + C) OSGDecoder, C >= 3")
    # rows 0 - 3 = sigma, r, g, b; further colour rows (decoder_output_dim 32) only feed the SR module no released sampler builds
    return tuple(t.detach().to(device, torch.float32).contiguous() for t in (n[0].weight, n[0].bias, n[2].weight[:4], n[2].bias[:4]))


class ImportanceRenderer(nn.Module):
    def __init__(self):
        super().__init__()

    @torch.no_grad()
    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options, return_meta=False,
                jitter=None, u_fine=None, planes_channel_last=None, plane_index=None, decoder_weights_dev=None):
        """planes [N, 3, C, H, W] (or [N, 3*C, H, W]); ray_origins / ray_directions [N, M, 3], any M.
        Alternatively planes_channel_last [NP, 3, H, W, 32] + plane_index [N] (many ray bundles over few tri-planes).
        Returns the reference's dict (renderer.py:276-300): feature_samples [N, M, 3], depth_samples [N, M, 1], weights_samples [N, M, 1],
        visibility [N, M, 1], shape_synthesized; with return_meta also weights [N, M, S + NI - 1, 1], all_coords [N, M, S + NI, 3],
        feature_volume [N, M, S + NI, 3] (return_meta needs rendering_options['return_sampling_details_flag'] in the reference - it
        reads shape_synthesized['coarse_coords'] - and does not here)."""
        rk = rendering_options
        check_rendering_options(rk)
        if not ray_origins.is_cuda:
            raise RuntimeError("ln3diff_amd.ImportanceRenderer runs on the HIP device only (no CPU fallback)")
        dev = ray_origins.device
        N, M, _ = ray_origins.shape
        if planes_channel_last is None:
            p4 = planes.reshape(planes.shape[0], -1, planes.shape[-2], planes.shape[-1])
            if p4.shape[1] != 96:
                raise NotImplementedError("3 planes x 32 channels only (no background tri-plane)")
            planes_channel_last = torch.empty(p4.shape[0], 3, p4.shape[2], p4.shape[3], 32, device=dev, dtype=torch.float32)
            ops.planes_to_channel_last(p4.contiguous().float(), planes_channel_last, p4.shape[0], 32, p4.shape[2], p4.shape[3])
            plane_index = torch.arange(N, device=dev, dtype=torch.int32)
        H, W = planes_channel_last.shape[2], planes_channel_last.shape[3]
        S, NI = rk.get('depth_resolution', 64), rk.get('depth_resolution_importance', 64)
        if jitter is None:
            jitter, u_fine = draw_render_noise(N, M, S, device=dev, n_importance=NI)
        jitter = jitter.to(dev, torch.float32).reshape(N, M, S).contiguous()
        u_fine = u_fine.to(dev, torch.float32).reshape(N * M, NI).contiguous()
        dec = decoder_weights_dev if decoder_weights_dev is not None else decoder_weights(decoder, dev)
        rgb = torch.empty(N, 3, M, device=dev)
        depth = torch.empty(N, M, 1, device=dev)
        wsum = torch.empty(N, M, 1, device=dev)
        vis = torch.empty(N, M, 1, device=dev)
        lim = torch.empty(N * M * 2, device=dev)
        scal = torch.empty(RENDER_SCRATCH_FLOATS, device=dev)
        details = bool(rk.get('return_sampling_details_flag', False))
        cs = torch.empty(N, M, S, device=dev) if details else None
        fs = torch.empty(N, M, NI, device=dev) if details else None
        cc = torch.empty(N, M, S, 3, device=dev) if details else None
        fc = torch.empty(N, M, NI, 3, device=dev) if details else None
        wall = torch.empty(N, M, S + NI - 1, 1, device=dev) if return_meta else None
        call = torch.empty(N, M, S + NI, 3, device=dev) if return_meta else None
        fvol = torch.empty(N, M, S + NI, 3, device=dev) if return_meta else None
        ops.render_triplane(planes_channel_last, H, W, plane_index.to(dev, torch.int32).contiguous(), None, 0, dec, jitter, u_fine,
                            rgb, depth, wsum, lim, scal, coarse_sigma=cs,
                            ray_o=ray_origins.to(torch.float32).contiguous(), ray_d=ray_directions.to(torch.float32).contiguous(),
                            fine_sigma=fs, coarse_coords=cc, fine_coords=fc, n_views=N, rays_per_view=M, visibility=vis,
                            weights=wall, all_coords=call, feature_volume=fvol, **render_call_kwargs(rk))
        ret = {'feature_samples': rgb.permute(0, 2, 1), 'depth_samples': depth, 'weights_samples': wsum,
               'shape_synthesized': {'depth': depth}, 'visibility': vis}
        if return_meta:
            ret.update(all_coords=call, feature_volume=fvol, weights=wall)
        if details:
            ret['shape_synthesized'].update(coarse_coords=cc, coarse_densities=cs.unsqueeze(-1), fine_coords=fc.reshape(N, M * NI, 3),
                                            fine_densities=fs.unsqueeze(-1))
        return ret
