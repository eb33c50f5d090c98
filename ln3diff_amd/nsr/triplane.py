"""Triplane renderer module on the fused HIP ray-marcher.

Mirrors the reference's nsr/triplane.py `Triplane` (forward(planes, c) -> dict with image_raw,
image_depth, weights_samples, image_mask, ...) and `OSGDecoder` (state-dict keys
`decoder.net.{0,2}.{weight,bias}`).  rendering_kwargs: any sampling preset of nsr/script_util.py:433-1000
(volumetric_rendering/renderer.py lists what is taken); the default is the Objaverse preset (:761-798: 64 + 64
samples, box_warp 0.9, bbox +-0.45, white background, auto ray limits).  ray generation, sampling, gather, MLP
and compositing all run inside `ln3d_render_triplane` (csrc/render.hip) - nothing of the reference's
[V,3,M*S,32] feature tensor or its sort/gather intermediates is ever materialised.
"""
import torch
import torch.nn as nn

from .. import ops, _cache
from .._lib import RENDER_SCRATCH_FLOATS, RENDER_MAX_CALLS
from .volumetric_rendering.renderer import ImportanceRenderer, check_rendering_options, draw_render_noise, render_call_kwargs  # noqa: F401


class FullyConnectedLayer(nn.Module):      # nsr/networks_stylegan2.py:122-157 (container; gain applied in-kernel)
    def __init__(self, in_features, out_features, lr_multiplier=1):
        super().__init__()
        self.weight = nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = nn.Parameter(torch.zeros([out_features]))


class OSGDecoder(nn.Module):               # nsr/triplane.py:339-372
    """32 -> 64 -> 1 + decoder_output_dim.  The Objaverse configurations use decoder_output_dim = 3; the ShapeNet / FFHQ launchers 32
    (`--decoder_output_dim 32`): their extra 29 colour channels only feed the super-resolution module, which no released sampler
    instantiates (every launcher passes --sr_training False, so sr_kwargs = {} and Triplane.superresolution is None:
    scripts/vit_triplane_diffusion_sample.py:83-90, nsr/triplane.py:476-500) - `image_raw = feature_image[:, :3]` (nsr/triplane.py:683).
    The kernels therefore evaluate output rows 0 - 3 (sigma, r, g, b) of whatever width the module holds."""

    def __init__(self, n_features=32, options=None):
        super().__init__()
        options = options or {'decoder_lr_mul': 1, 'decoder_output_dim': 3}
        assert options.get('decoder_lr_mul', 1) == 1
        self.hidden_dim = 64
        self.decoder_output_dim = options['decoder_output_dim']
        assert n_features == 32 and self.decoder_output_dim >= 3, "HIP renderer: 32 tri-plane features, at least the 3 colour outputs"
        self.net = nn.Sequential(FullyConnectedLayer(n_features, self.hidden_dim), nn.Softplus(),
                                 FullyConnectedLayer(self.hidden_dim, 1 + self.decoder_output_dim))


# nsr/script_util.py:433-465,761-798 (probe of rendering_options_defaults).  return_sampling_details_flag is
# True in the reference's preset; it defaults to False here because the sampling drivers never read `shape_synthesized` and the
# per-sample tensors cost 100 MB of HBM writes per 256^2 view - set it to get them.
OBJAVERSE_RENDERING_KWARGS = dict(
    depth_resolution=64, depth_resolution_importance=64, ray_start='auto', ray_end='auto', box_warp=0.9,
    white_back=True, sampler_bbox_min=-0.45, sampler_bbox_max=0.45, filter_out_of_bbox=True,
    clamp_mode='softplus', disparity_space_sampling=False, PatchRaySampler=True, decoder_lr_mul=1,
    return_sampling_details_flag=False, image_resolution=256, z_near=1.05, z_far=2.45, radius_range=[1.5, 2])


class Triplane(nn.Module):
    def __init__(self, c_dim=25, img_resolution=128, img_channels=3, out_chans=96, triplane_size=224,
                 rendering_kwargs=None, decoder_in_chans=32, decoder_output_dim=3, sr_kwargs=None, bcg_synthesis_kwargs=None,
                 lrm_decoder=False, create_triplane=False, **_):
        super().__init__()
        # The released sampling configurations build the renderer with sr_kwargs={} / no background model / the OSG decoder with
        # 3 colour channels (nsr/script_util.py:1149,1357-1369): the variants below change what the ray marcher composites (32
        # feature channels into a super-resolution CNN, nsr/triplane.py:476-500,695-711; a background NeRF; the LRM decoder) and
        # are not built - they are refused here rather than silently rendering something else.
        if sr_kwargs or bcg_synthesis_kwargs or lrm_decoder or create_triplane or decoder_output_dim < 3:
            raise NotImplementedError(
                "Triplane: sr_kwargs / bcg_synthesis_kwargs / lrm_decoder / create_triplane are outside the sampling hot path (every "
                "released sampler passes --sr_training False -> sr_kwargs={}, no background model, the OSG decoder); got "
                f"sr_kwargs={sr_kwargs!r}, bcg_synthesis_kwargs={bcg_synthesis_kwargs!r}, lrm_decoder={lrm_decoder}, "
                f"create_triplane={create_triplane}, decoder_output_dim={decoder_output_dim}")
        self.superresolution = None                      # attribute of the reference class (nsr/triplane.py:500)
        self.rendering_kwargs = dict(OBJAVERSE_RENDERING_KWARGS if rendering_kwargs is None else rendering_kwargs)
        check_rendering_options(self.rendering_kwargs)
        self.renderer = ImportanceRenderer()             # the explicit-ray seam (nsr/triplane.py:470 in the reference)
        self.neural_rendering_resolution = img_resolution
        self.decoder_in_chans = decoder_in_chans
        self.decoder = OSGDecoder(decoder_in_chans, {'decoder_lr_mul': 1, 'decoder_output_dim': decoder_output_dim})
        self._dec = None
        self._dec_epoch = -1
        _cache.watch(self)

    def _apply(self, fn, *a, **k):
        _cache.bump()
        return super()._apply(fn, *a, **k)

    def _decoder_dev(self, dev):
        if self._dec is None or self._dec[0].device != dev or self._dec_epoch != _cache.EPOCH[0]:
            self._dec_epoch = _cache.EPOCH[0]
            _cache.watch_tree(self)
            n = self.decoder.net                        # output rows 0 - 3 = sigma, r, g, b (a wider decoder's other rows feed SR only)
            self._dec = tuple(t.detach().to(dev, torch.float32).contiguous()
                              for t in (n[0].weight, n[0].bias, n[2].weight[:4], n[2].bias[:4]))
        return self._dec

    @staticmethod
    def to_channel_last(planes):
        """[NP, 96, H, W] (reference '(n c) h w') -> [NP, 3, H, W, 32] f32 for the gather kernel."""
        NP, C3, H, W = planes.shape
        out = torch.empty(NP, 3, H, W, C3 // 3, device=planes.device, dtype=torch.float32)
        ops.planes_to_channel_last(planes.contiguous().float(), out, NP, C3 // 3, H, W)
        return out

    @torch.no_grad()
    def forward(self, planes=None, c=None, neural_rendering_resolution=None, jitter=None, u_fine=None,
                planes_channel_last=None, plane_index=None, return_debug=False, views_per_call=0, **_):
        """planes [V,96,H,W] (one tri-plane per camera row, as the reference) or
        planes_channel_last [NP,3,H,W,32] + plane_index [V] (many views of few tri-planes).
        views_per_call: the reference reduces the ray-limit fix-up and the depth clamp range over everything ONE forward() call
        renders; 0 = this call is one such call (reference semantics of Triplane.forward), k = every k consecutive views are
        (the drivers, which call the reference once per camera, pass 1)."""
        res = neural_rendering_resolution or self.neural_rendering_resolution
        self.neural_rendering_resolution = res
        if not c.is_cuda:
            raise RuntimeError("ln3diff_amd.Triplane runs on the HIP device only (no CPU fallback)")
        dev = c.device
        rk = self.rendering_kwargs
        V, M, S, NI = c.shape[0], res * res, rk.get('depth_resolution', 64), rk.get('depth_resolution_importance', 64)
        if planes_channel_last is None:
            planes_channel_last = self.to_channel_last(planes)
            plane_index = torch.arange(V, device=dev, dtype=torch.int32)
        H, W = planes_channel_last.shape[2], planes_channel_last.shape[3]
        if jitter is None:
            jitter, u_fine = draw_render_noise(V, M, S, device=dev, n_importance=NI)
        jitter = jitter.to(dev, torch.float32).reshape(V, M, S).contiguous()
        u_fine = u_fine.to(dev, torch.float32).reshape(V * M, NI).contiguous()
        rgb = torch.empty(V, 3, res, res, device=dev)
        depth = torch.empty(V, 1, res, res, device=dev)
        wsum = torch.empty(V, 1, res, res, device=dev)
        lim = torch.empty(V * M * 2, device=dev)
        scal = torch.empty(RENDER_SCRATCH_FLOATS, device=dev)
        # sampling details (the reference's `shape_synthesized`, nsr/triplane.py:569-573 / renderer.py:196-283): 100 MB per 256^2
        # view, so only on request - rendering_kwargs['return_sampling_details_flag'] (the reference preset sets it) or return_debug
        details = bool(rk.get('return_sampling_details_flag', False))
        cs = torch.empty(V, M, S, device=dev) if (return_debug or details) else None
        fd = torch.empty(V, M, NI, device=dev) if return_debug else None
        fs = torch.empty(V, M, NI, device=dev) if details else None
        cc = torch.empty(V, M, S, 3, device=dev) if details else None
        fc = torch.empty(V, M, NI, 3, device=dev) if details else None
        pidx = plane_index.to(dev, torch.int32).contiguous()
        cam = c.to(torch.float32).contiguous()
        # one launch handles at most RENDER_MAX_CALLS reference calls (range records in the scratch): more are rendered in chunks of
        # whole calls, which is exact because calls are independent
        vpc = V if (views_per_call <= 0 or views_per_call > V) else views_per_call
        chunk = V if (V + vpc - 1) // vpc <= RENDER_MAX_CALLS else RENDER_MAX_CALLS * vpc
        sl = lambda t, a, b, per=1: None if t is None else t[a * per:b * per]
        for a in range(0, V, chunk):
            b = min(V, a + chunk)
            ops.render_triplane(planes_channel_last, H, W, pidx[a:b], cam[a:b], res, self._decoder_dev(dev), jitter[a:b], u_fine[a * M:b * M],
                                rgb[a:b], depth[a:b], wsum[a:b], lim, scal, coarse_sigma=sl(cs, a, b),
                                fine_depths=sl(fd, a, b), fine_sigma=sl(fs, a, b), coarse_coords=sl(cc, a, b), fine_coords=sl(fc, a, b),
                                views_per_call=views_per_call if chunk == V else vpc, **render_call_kwargs(rk))
        ret = {'feature_image': rgb, 'image_raw': rgb, 'image_depth': depth, 'weights_samples': wsum,
               'image_mask': wsum * (1 + 2 * 0.001) - 0.001,
               'shape_synthesized': {'image_depth': depth, 'depth': depth.reshape(V, M, 1)}}
        if return_debug:
            ret['shape_synthesized'].update(coarse_densities=cs.unsqueeze(-1), fine_depths=fd.unsqueeze(-1))
        if details:
            ret['shape_synthesized'].update(coarse_coords=cc, coarse_densities=cs.unsqueeze(-1), fine_coords=fc.reshape(V, M * NI, 3),
                                            fine_densities=fs.unsqueeze(-1))
        return ret

    @torch.no_grad()
    def query_points(self, planes_channel_last_one, points):
        """points [P,3] against ONE tri-plane [3,H,W,32] -> {'sigma':[P,1],'rgb':[P,3]} (no bbox filter:
        renderer._run_model as used by forward_points, vit/vit_triplane.py:2026-2041)."""
        dev = points.device
        P = points.shape[0]
        H, W = planes_channel_last_one.shape[-3], planes_channel_last_one.shape[-2]
        sigma = torch.empty(P, 1, device=dev)
        rgb = torch.empty(P, 3, device=dev)
        scal = torch.empty(RENDER_SCRATCH_FLOATS, device=dev)
        ops.query_points(planes_channel_last_one.contiguous(), H, W, points.contiguous().float(), self._decoder_dev(dev),
                         self.rendering_kwargs['box_warp'], sigma, rgb, scal)
        return {'sigma': sigma, 'rgb': rgb}
