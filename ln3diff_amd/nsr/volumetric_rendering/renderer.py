"""`ImportanceRenderer` - the lower seam of the reference's renderer (nsr/volumetric_rendering/renderer.py:125-307) on the fused
HIP ray-marcher: `forward(planes, decoder, ray_origins, ray_directions, rendering_options, return_meta)` with explicit rays.

Everything the reference does between this call and its return - AABB ray limits (math_utils.get_ray_limits_box), the invalid-ray
fix-up, stratified sampling, `sample_from_planes` x3, the decoder MLP, MipRayMarcher2, importance sampling, `unify_samples`
and the second ray-march - runs inside `ln3d_render_triplane` (csrc/render.hip) in its explicit-ray mode.  Random numbers are
inputs (`jitter` [N, M, 64] for `sample_stratified`, `u_fine` [N*M, 64] for `sample_pdf`); `draw_render_noise` restates the
reference's RNG consumption order.

Supported rendering_options: the Objaverse preset family (nsr/script_util.py:761-798) - ray_start = ray_end = 'auto',
depth_resolution = depth_resolution_importance = 64, clamp_mode 'softplus', filter_out_of_bbox, white_back on or off.
Not produced (consumers are training-time only): 'visibility', and the `return_meta` extras 'all_coords' / 'feature_volume' /
'weights' (the merged, sorted per-sample tensors never exist in memory here).
"""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import RENDER_SCRATCH_FLOATS


def draw_render_noise(V, M, S=64, generator=None, device='cpu'):
    """The reference's RNG consumption per Triplane.forward, as logical tensors:
    coarse jitter = rand_like on a [S,V,M,1]-strided tensor, then fine uniforms rand(V*M, S)."""
    if device == 'cpu' or str(device) == 'cpu':
        j = torch.rand(S, V, M, 1, generator=generator).permute(1, 2, 0, 3).reshape(V, M, S).contiguous()
        u = torch.rand(V * M, S, generator=generator)
    else:
        j = torch.rand(V, M, S, device=device, generator=generator)
        u = torch.rand(V * M, S, device=device, generator=generator)
    return j, u


def check_rendering_options(rk):
    if rk.get('depth_resolution', 64) != 64 or rk.get('depth_resolution_importance', 64) != 64:
        raise NotImplementedError("the HIP renderer is built for 64 coarse + 64 importance samples per ray (Objaverse preset)")
    if not (rk.get('ray_start') == rk.get('ray_end') == 'auto'):
        raise NotImplementedError("ray_start / ray_end must be 'auto' (ray / AABB limits)")
    if not rk.get('filter_out_of_bbox', False) or rk.get('clamp_mode', 'softplus') != 'softplus':
        raise NotImplementedError("filter_out_of_bbox=True and clamp_mode='softplus' only")
    if rk.get('disparity_space_sampling', False) or rk.get('density_noise', 0) > 0:
        raise NotImplementedError("disparity-space sampling / density noise are not part of the sampling hot path")


def decoder_weights(decoder, device):
    """(w0, b0, w1, b1) of an OSGDecoder-shaped module (`net.0`, `net.2` FullyConnectedLayers: 32 -> 64 -> 1 + 3)."""
    n = decoder.net
    w = tuple(t.detach().to(device, torch.float32).contiguous() for t in (n[0].weight, n[0].bias, n[2].weight, n[2].bias))
    if tuple(w[0].shape) != (64, 32) or tuple(w[2].shape) != (4, 64):
        raise NotImplementedError("the HIP renderer is built for the released 32 -> 64 -> 4 OSGDecoder")
    return w


class ImportanceRenderer(nn.Module):
    def __init__(self):
        super().__init__()

    @torch.no_grad()
    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options, return_meta=False,
                jitter=None, u_fine=None, planes_channel_last=None, plane_index=None, decoder_weights_dev=None):
        """planes [N, 3, C, H, W] (or [N, 3*C, H, W]); ray_origins / ray_directions [N, M, 3], M a square number of rays.
        Alternatively planes_channel_last [NP, 3, H, W, 32] + plane_index [N] (many ray bundles over few tri-planes)."""
        rk = rendering_options
        check_rendering_options(rk)
        if not ray_origins.is_cuda:
            raise RuntimeError("ln3diff_amd.ImportanceRenderer runs on the HIP device only (no CPU fallback)")
        dev = ray_origins.device
        N, M, _ = ray_origins.shape
        res = int(round(M ** 0.5))
        if res * res != M:
            raise NotImplementedError("M must be a square number of rays (one res x res bundle per batch element)")
        if planes_channel_last is None:
            p4 = planes.reshape(planes.shape[0], -1, planes.shape[-2], planes.shape[-1])
            if p4.shape[1] != 96:
                raise NotImplementedError("3 planes x 32 channels only (no background tri-plane)")
            planes_channel_last = torch.empty(p4.shape[0], 3, p4.shape[2], p4.shape[3], 32, device=dev, dtype=torch.float32)
            ops.planes_to_channel_last(p4.contiguous().float(), planes_channel_last, p4.shape[0], 32, p4.shape[2], p4.shape[3])
            plane_index = torch.arange(N, device=dev, dtype=torch.int32)
        H, W = planes_channel_last.shape[2], planes_channel_last.shape[3]
        S = 64
        if jitter is None:
            jitter, u_fine = draw_render_noise(N, M, S, device=dev)
        jitter = jitter.to(dev, torch.float32).reshape(N, M, S).contiguous()
        u_fine = u_fine.to(dev, torch.float32).reshape(N * M, S).contiguous()
        dec = decoder_weights_dev if decoder_weights_dev is not None else decoder_weights(decoder, dev)
        rgb = torch.empty(N, 3, res, res, device=dev)
        depth = torch.empty(N, 1, res, res, device=dev)
        wsum = torch.empty(N, 1, res, res, device=dev)
        lim = torch.empty(N * M * 2, device=dev)
        scal = torch.empty(RENDER_SCRATCH_FLOATS, device=dev)
        details = bool(rk.get('return_sampling_details_flag', False))
        cs = torch.empty(N, M, S, device=dev) if details else None
        fs = torch.empty(N, M, S, device=dev) if details else None
        cc = torch.empty(N, M, S, 3, device=dev) if details else None
        fc = torch.empty(N, M, S, 3, device=dev) if details else None
        ops.render_triplane(planes_channel_last, H, W, plane_index.to(dev, torch.int32).contiguous(), None, res, dec, jitter, u_fine,
                            rgb, depth, wsum, lim, scal, box_warp=rk['box_warp'], bbox_min=rk['sampler_bbox_min'],
                            bbox_max=rk['sampler_bbox_max'], white_back=rk.get('white_back', False), coarse_sigma=cs,
                            ray_o=ray_origins.to(torch.float32).contiguous(), ray_d=ray_directions.to(torch.float32).contiguous(),
                            fine_sigma=fs, coarse_coords=cc, fine_coords=fc, n_views=N)
        ret = {'feature_samples': rgb.reshape(N, 3, M).permute(0, 2, 1), 'depth_samples': depth.reshape(N, M, 1),
               'weights_samples': wsum.reshape(N, M, 1), 'shape_synthesized': {'depth': depth.reshape(N, M, 1)}}
        if details:
            ret['shape_synthesized'].update(coarse_coords=cc, coarse_densities=cs.unsqueeze(-1), fine_coords=fc.reshape(N, M * S, 3),
                                            fine_densities=fs.unsqueeze(-1))
        return ret
