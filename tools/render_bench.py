#!/usr/bin/env python
"""Timing of the fused tri-plane ray-marcher at the benchmarked resolutions (GPU box).  LN3D_LIB selects an alternative build of
the library (bench-only ablation builds: -DLN3D_RENDER_ABL=n, see csrc/render.hip)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LN3D_LIB'):
    from ln3diff_amd import _lib
    _lib.LIB_PATH = os.environ['LN3D_LIB']
from ln3diff_amd.nsr.triplane import Triplane          # noqa: E402
from ln3diff_amd.synth import orbit_cameras            # noqa: E402

from ln3diff_amd.nsr.triplane import OBJAVERSE_RENDERING_KWARGS  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
# RENDER_PRESET (r6): the presets that take render_generic_kernel - never timed before.  objv128 = 128 + 128 samples, auto
# limits; shapenet = 64 + 64 samples with the ShapeNet launchers' numeric --ray_start 0.6 --ray_end 1.8; eg3d48 = 48 + 48
PRESETS = {'objv128': dict(OBJAVERSE_RENDERING_KWARGS, depth_resolution=128, depth_resolution_importance=128),
           'shapenet': dict(OBJAVERSE_RENDERING_KWARGS, ray_start=0.6, ray_end=1.8, box_warp=1.0, sampler_bbox_min=-0.5, sampler_bbox_max=0.5),
           'eg3d48': dict(OBJAVERSE_RENDERING_KWARGS, depth_resolution=48, depth_resolution_importance=48)}
preset = os.environ.get('RENDER_PRESET')
NS = 64 if not preset else PRESETS[preset]['depth_resolution']
tp = Triplane(img_resolution=256, rendering_kwargs=PRESETS[preset] if preset else None).to(dev)
tp.decoder.net[2].bias.data[0] += 4.0
pcl = torch.randn(1, 3, 128, 128, 32, device=dev) * 4
CASES = ((256, 4), (128, 8), (512, 2))
if len(sys.argv) > 1:
    CASES = tuple(c for c in CASES if c[0] == int(sys.argv[1]))
for res, V in CASES:
    cams = orbit_cameras(V).to(dev)
    idx = torch.zeros(V, dtype=torch.int32, device=dev)
    j = torch.rand(V, res * res, NS, device=dev)
    u = torch.rand(V * res * res, PRESETS[preset]['depth_resolution_importance'] if preset else 64, device=dev)
    f = lambda: tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
    out = f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(3):
            f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 3)
    img = out['image_raw']
    print(f'[{preset or "objaverse 64+64 (render_kernel)"}] {res}^2 x {V} views: {best / V:7.3f} ms/view   ({V * res * res * 2 * NS / best / 1e6:7.2f} G sample points/s)   '
          f'finite {bool(torch.isfinite(img).all())} mean {float(img.mean()):+.4f} mask {float(out["image_mask"].mean()):.3f}')
