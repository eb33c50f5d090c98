#!/usr/bin/env python
"""Bitwise repeatability of the SHIPPED ray-marcher over more scenes than the test suite's one (profiles/r4_render_spill.md: the
irreproducible rays of the SLP build are few and scene-dependent, so one scene passing says little): random planes, several orbits
(elevation / radius / view count), 128^2 - 512^2, every configuration rendered 4 times."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LN3D_LIB'):
    from ln3diff_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['LN3D_LIB'])
from ln3diff_amd.nsr.triplane import Triplane          # noqa: E402
from ln3diff_amd.synth import synth_input, orbit_cameras            # noqa: E402

tp = Triplane(img_resolution=256).cuda()
tp.decoder.net[2].bias.data[0] += 4.0
bad = total = 0
for seed, (V, res, el, rad) in enumerate(((4, 256, 15.0, 1.7719), (8, 256, 40.0, 1.5), (12, 128, -20.0, 2.2), (2, 512, 5.0, 1.7719),
                                          (8, 256, 75.0, 1.3), (24, 256, 0.0, 1.9))):
    planes = synth_input('planes', (1, 96, 128, 128), 10 + seed, 4.0).cuda()
    pcl = tp.to_channel_last(planes)
    cams = orbit_cameras(V, radius=rad, elevation_deg=el).cuda()
    g = torch.Generator(device='cuda').manual_seed(seed)
    j = torch.rand(V, res * res, 64, device='cuda', generator=g)
    u = torch.rand(V * res * res, 64, device='cuda', generator=g)
    idx = torch.zeros(V, dtype=torch.int32, device='cuda')
    f = lambda: tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
    ref = f()
    nd = 0
    for rep in range(3):
        o = f()
        nd += sum(int((ref[k] != o[k]).sum()) for k in ('image_raw', 'image_depth', 'weights_samples'))
    total += 1
    bad += nd > 0
    print(f'{V:3d} views @ {res}^2, elevation {el:+.0f}, radius {rad}: {nd} differing values over 3 repeats; mask mean {float(ref["image_mask"].mean()):.3f}')
print('REPEATABLE' if bad == 0 else f'{bad} of {total} configurations differ')
