"""CPU oracle (TEST INFRASTRUCTURE ONLY) - fp32 torch restatement of the tri-plane
volumetric renderer on the sampling path.  Never imported by the product path.

Reference citations:
  ray generation      nsr/volumetric_rendering/ray_sampler.py:262-331 (PatchRaySampler with
                      patch == resolution == full image), create_patch_uv :66-181
  ray/AABB limits     nsr/volumetric_rendering/math_utils.py:46-118 (get_ray_limits_box)
  linspace            nsr/volumetric_rendering/math_utils.py:121-137
  renderer            nsr/volumetric_rendering/renderer.py:133-307 (forward),
                      :55-104 (project_onto_planes / sample_from_planes),
                      :354-407 (_forward_pass bbox filter), :437-477 (sample_stratified),
                      :479-552 (sample_importance / sample_pdf), :422-435 (unify_samples)
  OSGDecoder          nsr/triplane.py:339-372, FullyConnectedLayer nsr/networks_stylegan2.py:122-157
  ray marcher         nsr/volumetric_rendering/ray_marcher.py:26-68
  Triplane.forward    nsr/triplane.py:505-750 (image assembly, image_mask)
  grid query          vit/vit_triplane.py:2009-2112 (forward_points / triplane_decode_grid)

Random draws are explicit inputs (`jitter` [V,M,S,1] logical layout and `u_fine` [V*M,Nimp]);
see SURVEY.md App. A.13 for how the reference's RNG stream maps onto them.
"""
import math

import torch
import torch.nn.functional as F

OBJAVERSE_OPTS = dict(  # nsr/script_util.py:761-798 as probed in SURVEY.md App. A.14
    depth_resolution=64, depth_resolution_importance=64, box_warp=0.9, white_back=True,
    sampler_bbox_min=-0.45, sampler_bbox_max=0.45, filter_out_of_bbox=True,
    clamp_mode='softplus', ray_start='auto', ray_end='auto', disparity_space_sampling=False)


SHAPENET_OPTS = dict(  # nsr/script_util.py 'shapenet_tuneray_aug_resolution_64_64_nearestSR' with the released --ray_start 0.6 --ray_end 1.8
    depth_resolution=64, depth_resolution_importance=64, ray_start=0.6, ray_end=1.8, box_warp=1.2, white_back=True,
    clamp_mode='softplus', disparity_space_sampling=False)
OBJAVERSE_128_OPTS = dict(OBJAVERSE_OPTS, depth_resolution=128, depth_resolution_importance=128)   # 'objverse_tuneray_aug_resolution_128_128_auto'
OBJAVERSE_96_OPTS = dict(OBJAVERSE_OPTS, depth_resolution=96, depth_resolution_importance=96)      # '..._96_96_auto'
EG3D_80_OPTS = dict(  # 'eg3d_shapenet_aug_resolution': 80 + 80, numeric limits, box_warp 1.1
    depth_resolution=80, depth_resolution_importance=80, ray_start=0.1, ray_end=1.9, box_warp=1.1, white_back=True,
    clamp_mode='softplus', disparity_space_sampling=False)
AFHQ_48_OPTS = dict(  # 'afhq' / 'ffhq': 48 + 48, ray_start 2.25, ray_end 3.3, box_warp 1, black background
    depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1, white_back=False,
    clamp_mode='softplus', disparity_space_sampling=False)


def make_rays(c, res):
    """c [V,25] = 16 cam2world + 9 intrinsics -> ray_o, ray_d [V, res*res, 3]."""
    V = c.shape[0]
    c2w = c[:, :16].reshape(V, 4, 4)
    K = c[:, 16:25].reshape(V, 3, 3)
    fx, fy, cx, cy, sk = K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2], K[:, 0, 1]
    ar = torch.arange(res, dtype=torch.float32)
    ii, jj = torch.meshgrid(ar, ar, indexing='ij')
    uv = torch.stack([ii, jj]) * (1. / res) + (0.5 / res)
    uv = uv.flip(0).reshape(2, -1).transpose(1, 0)            # [M,2] = (x=col, y=row)
    x_cam = uv[None, :, 0].expand(V, -1)
    y_cam = uv[None, :, 1].expand(V, -1)
    z_cam = torch.ones_like(x_cam)
    u = lambda t: t.unsqueeze(-1)
    x_lift = (x_cam - u(cx) + u(cy) * u(sk) / u(fy) - u(sk) * y_cam / u(fy)) / u(fx) * z_cam
    y_lift = (y_cam - u(cy)) / u(fy) * z_cam
    pts = torch.stack((x_lift, y_lift, z_cam, torch.ones_like(z_cam)), dim=-1)
    world = torch.bmm(c2w, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    cam = c2w[:, :3, 3]
    d = F.normalize(world - cam[:, None, :], dim=2)
    o = cam.unsqueeze(1).repeat(1, d.shape[1], 1)
    return o, d


def ray_limits_box(rays_o, rays_d, box_side_length):
    shp = rays_o.shape
    o, d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    half = box_side_length / 2
    bounds = torch.tensor([[-half] * 3, [half] * 3], dtype=o.dtype)
    valid = torch.ones(o.shape[0], dtype=torch.bool)
    inv = 1 / d
    sign = (inv < 0).long()
    sel = lambda s, ax: bounds.index_select(0, s[:, ax])[:, ax]
    tmin = (sel(sign, 0) - o[:, 0]) * inv[:, 0]
    tmax = (sel(1 - sign, 0) - o[:, 0]) * inv[:, 0]
    tymin = (sel(sign, 1) - o[:, 1]) * inv[:, 1]
    tymax = (sel(1 - sign, 1) - o[:, 1]) * inv[:, 1]
    valid[torch.logical_or(tmin > tymax, tymin > tmax)] = False
    tmin, tmax = torch.max(tmin, tymin), torch.min(tmax, tymax)
    tzmin = (sel(sign, 2) - o[:, 2]) * inv[:, 2]
    tzmax = (sel(1 - sign, 2) - o[:, 2]) * inv[:, 2]
    valid[torch.logical_or(tmin > tzmax, tzmin > tmax)] = False
    tmin, tmax = torch.max(tmin, tzmin), torch.min(tmax, tzmax)
    tmin = torch.where(valid, tmin, torch.full_like(tmin, -1.))
    tmax = torch.where(valid, tmax, torch.full_like(tmax, -2.))
    return tmin.reshape(*shp[:-1], 1), tmax.reshape(*shp[:-1], 1)


def auto_ray_range(ray_o, ray_d, box_warp):
    """renderer.py:145-155 including the invalid-ray fix-up (sic: both from ray_start)."""
    start, end = ray_limits_box(ray_o, ray_d, box_warp)
    ok = end > start
    if torch.any(ok):
        smin, smax = start[ok].min(), start[ok].max()
        start = torch.where(ok, start, smin)
        end = torch.where(ok, end, smax)
    return start, end


def stratified_depths(start, end, S, jitter):
    """start,end [V,M,1]; jitter [V,M,S,1] in [0,1) -> depths [V,M,S,1]."""
    steps = (torch.arange(S, dtype=torch.float32) / (S - 1)).view(1, 1, S, 1)
    d = start[:, :, None] + steps * (end - start)[:, :, None]
    delta = (end - start) / (S - 1)
    return d + jitter * delta[..., None]


def bilinear_zeros(plane, gx, gy):
    """F.grid_sample(bilinear, zeros, align_corners=False) restated.
    plane [C,H,W]; gx,gy [P] in normalised coords -> [P,C]."""
    C, H, W = plane.shape
    ix = ((gx + 1) * W - 1) / 2
    iy = ((gy + 1) * H - 1) / 2
    x0, y0 = torch.floor(ix), torch.floor(iy)
    wx1, wy1 = ix - x0, iy - y0
    out = torch.zeros(gx.shape[0], C, dtype=plane.dtype)
    flat = plane.reshape(C, H * W)
    for dy, wy in ((0, 1 - wy1), (1, wy1)):
        for dx, wx in ((0, 1 - wx1), (1, wx1)):
            xi, yi = (x0 + dx).long(), (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1))
            v = flat[:, idx].t()
            out = out + v * (wx * wy * ok.float())[:, None]
    return out


def sample_planes(planes, coords, box_warp):
    """planes [N,3,C,H,W], coords [N,P,3] -> features [N,3,P,C] (sample_from_planes)."""
    N = planes.shape[0]
    cs = (2 / box_warp) * coords
    proj = ((0, 1), (1, 2), (2, 0))                           # (x,y) (y,z) (z,x)
    out = []
    for n in range(N):
        out.append(torch.stack([bilinear_zeros(planes[n, k], cs[n, :, a], cs[n, :, b])
                                for k, (a, b) in enumerate(proj)]))
    return torch.stack(out)


def osg_decoder(sd, feats, prefix='net.'):
    """feats [N,3,P,C] -> rgb [N,P,3], sigma [N,P,1]."""
    x = feats.mean(1)
    w0, b0 = sd[prefix + '0.weight'], sd[prefix + '0.bias']
    w1, b1 = sd[prefix + '2.weight'], sd[prefix + '2.bias']
    h = torch.addmm(b0.unsqueeze(0), x.reshape(-1, x.shape[-1]), (w0 * (1 / math.sqrt(w0.shape[1]))).t())
    h = F.softplus(h)
    y = torch.addmm(b1.unsqueeze(0), h, (w1 * (1 / math.sqrt(w1.shape[1]))).t())
    y = y.view(x.shape[0], x.shape[1], -1)
    return torch.sigmoid(y[..., 1:]) * (1 + 2 * 0.001) - 0.001, y[..., 0:1]


def run_model(planes, dec_sd, coords, opts, filter_bbox):
    rgb, sigma = osg_decoder(dec_sd, sample_planes(planes, coords, opts['box_warp']))
    if filter_bbox:
        inb = ((opts['sampler_bbox_min'] <= coords) & (coords <= opts['sampler_bbox_max'])).all(-1)
        fill = torch.nan_to_num(torch.full_like(sigma, -float('inf'))) / 3
        rgb = torch.where(inb[..., None], rgb, torch.zeros_like(rgb))
        sigma = torch.where(inb[..., None], sigma, fill)
    return rgb, sigma


def ray_march(colors, dens, depths, white_back=True):
    """MipRayMarcher2.run_forward. [V,M,S,*]."""
    deltas = depths[:, :, 1:] - depths[:, :, :-1]
    c_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / 2
    d_mid = (dens[:, :, :-1] + dens[:, :, 1:]) / 2
    z_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / 2
    d_mid = F.softplus(d_mid - 1)
    alpha = 1 - torch.exp(-(d_mid * deltas))
    shifted = torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2)
    T = torch.cumprod(shifted, -2)
    w = alpha * T[:, :, :-1]
    rgb = torch.sum(w * c_mid, -2)
    wt = w.sum(2)
    depth = torch.sum(w * z_mid, -2)
    depth = torch.nan_to_num(depth, float('inf'))
    depth = torch.clamp(depth, torch.min(depths), torch.max(depths))
    if white_back:
        rgb = rgb + 1 - wt
    return rgb * 2 - 1, depth, T[:, :, -1], w


def sample_importance(z_vals, weights, n_imp, u):
    V, M, S, _ = z_vals.shape
    z = z_vals.reshape(V * M, S)
    w = weights.reshape(V * M, -1)
    w = F.max_pool1d(w.unsqueeze(1).float(), 2, 1, padding=1)
    w = F.avg_pool1d(w, 2, 1).squeeze(1) + 0.01
    z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
    return sample_pdf(z_mid, w[:, 1:-1], n_imp, u).reshape(V, M, n_imp, 1)


def sample_pdf(bins, weights, n_imp, u, eps=1e-5):
    n_rays, n_s = weights.shape
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_s)
    idx = torch.stack([below, above], -1).view(n_rays, 2 * n_imp)
    cdf_g = torch.gather(cdf, 1, idx).view(n_rays, n_imp, 2)
    bins_g = torch.gather(bins, 1, idx).view(n_rays, n_imp, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def numeric_depths(ray_start, ray_end, S, jitter):
    """renderer.py:463-474 (ray_start / ray_end as numbers): torch.linspace + rand * delta.  jitter [V,M,S,1]."""
    z = torch.linspace(ray_start, ray_end, S).reshape(1, 1, S, 1)
    delta = (ray_end - ray_start) / (S - 1)
    return z + jitter * delta


def render(planes, dec_sd, ray_o, ray_d, jitter, u_fine, opts=OBJAVERSE_OPTS):
    """ImportanceRenderer.forward (renderer.py:133-307).  planes [V,3,C,H,W] (one tri-plane per view row).
    Also returns what `return_meta` adds (:283-300): weights [V,M,S+NI-1,1], all_coords [V,M,S+NI,3], feature_volume."""
    V, M, _ = ray_o.shape
    S, NI = opts['depth_resolution'], opts['depth_resolution_importance']
    if opts['ray_start'] == opts['ray_end'] == 'auto':
        start, end = auto_ray_range(ray_o, ray_d, opts['box_warp'])
        z_c = stratified_depths(start, end, S, jitter)
    else:
        z_c = numeric_depths(opts['ray_start'], opts['ray_end'], S, jitter).expand(V, M, S, 1).contiguous()
    pts = (ray_o.unsqueeze(-2) + z_c * ray_d.unsqueeze(-2)).reshape(V, -1, 3)
    fb = opts.get('filter_out_of_bbox', False)
    wb = opts.get('white_back', True)
    rgb_c, sig_c = run_model(planes, dec_sd, pts, opts, fb)
    rgb_c, sig_c = rgb_c.reshape(V, M, S, -1), sig_c.reshape(V, M, S, 1)
    out = dict(coarse_depths=z_c, coarse_densities=sig_c, coarse_colors=rgb_c, coarse_coords=pts.reshape(V, M, S, 3))
    if NI > 0:
        _, _, _, w_c = ray_march(rgb_c, sig_c, z_c, wb)
        z_f = sample_importance(z_c, w_c, NI, u_fine)
        pts_f = (ray_o.unsqueeze(-2) + z_f * ray_d.unsqueeze(-2)).reshape(V, -1, 3)
        rgb_f, sig_f = run_model(planes, dec_sd, pts_f, opts, fb)
        rgb_f, sig_f = rgb_f.reshape(V, M, NI, -1), sig_f.reshape(V, M, NI, 1)
        z_all = torch.cat([z_c, z_f], -2)
        rgb_all = torch.cat([rgb_c, rgb_f], -2)
        sig_all = torch.cat([sig_c, sig_f], -2)
        _, idx = torch.sort(z_all, dim=-2)
        z_all = torch.gather(z_all, -2, idx)
        rgb_all = torch.gather(rgb_all, -2, idx.expand(-1, -1, -1, rgb_all.shape[-1]))
        sig_all = torch.gather(sig_all, -2, idx)
        rgb, depth, vis, w = ray_march(rgb_all, sig_all, z_all, wb)
        coords_all = torch.gather(torch.cat([pts.reshape(V, M, S, 3), pts_f.reshape(V, M, NI, 3)], -2), -2, idx.expand(-1, -1, -1, 3))
        out.update(fine_depths=z_f, fine_densities=sig_f, coarse_weights=w_c, fine_coords=pts_f.reshape(V, M, NI, 3),
                   all_coords=coords_all, feature_volume=rgb_all, all_depths=z_all)
    else:
        rgb, depth, vis, w = ray_march(rgb_c, sig_c, z_c, wb)
    out.update(rgb=rgb, depth=depth, weights_sum=w.sum(2), visibility=vis, weights=w)
    return out


def triplane_render(planes96, dec_sd, c, res, jitter, u_fine, opts=OBJAVERSE_OPTS):
    """Triplane.forward: planes96 [V,96,H,W] -> image dict (image_raw in [-1,1])."""
    V = c.shape[0]
    planes = planes96.reshape(V, 3, -1, planes96.shape[-2], planes96.shape[-1])
    o, d = make_rays(c, res)
    r = render(planes, dec_sd, o, d, jitter, u_fine, opts)
    img = r['rgb'].permute(0, 2, 1).reshape(V, -1, res, res)
    depth = r['depth'].permute(0, 2, 1).reshape(V, 1, res, res)
    ws = r['weights_sum'].permute(0, 2, 1).reshape(V, 1, res, res)
    return dict(image_raw=img[:, :3], image_depth=depth, weights_samples=ws,
                image_mask=ws * (1 + 2 * 0.001) - 0.001, detail=r)


def decode_grid(planes96, dec_sd, grid_size, opts=OBJAVERSE_OPTS):
    """triplane_decode_grid: G^3 points over the sampler bbox, NO bbox filter, no march."""
    N = planes96.shape[0]
    planes = planes96.reshape(N, 3, -1, planes96.shape[-2], planes96.shape[-1])
    lo, hi = opts['sampler_bbox_min'], opts['sampler_bbox_max']
    ax = torch.linspace(lo, hi, grid_size)
    pts = torch.stack(torch.meshgrid(ax, ax, ax, indexing='ij'), dim=-1).reshape(1, -1, 3)
    pts = pts.expand(N, -1, -1)
    rgb, sigma = run_model(planes, dec_sd, pts, opts, False)
    g = grid_size
    return dict(rgb=rgb.reshape(N, g, g, g, 3), sigma=sigma.reshape(N, g, g, g, 1))
