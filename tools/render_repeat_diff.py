#!/usr/bin/env python
"""Where do two launches of the ray-marcher on the same inputs differ?  (LN3D_LIB selects the build: profiles/r4_render_spill.md)"""
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
planes = synth_input('planes', (1, 96, 128, 128), 3, 4.0).cuda()
pcl = tp.to_channel_last(planes)
V = 4
cams = orbit_cameras(V).cuda()
g = torch.Generator(device='cuda').manual_seed(0)
j = torch.rand(V, 256 * 256, 64, device='cuda', generator=g)
u = torch.rand(V * 256 * 256, 64, device='cuda', generator=g)
idx = torch.zeros(V, dtype=torch.int32, device='cuda')
DEBUG = len(sys.argv) > 1 and sys.argv[1] == 'debug'     # also fetch coarse densities / fine depths of the differing rays
f = lambda: tp(c=cams, planes_channel_last=pcl, plane_index=idx, jitter=j, u_fine=u, return_debug=DEBUG)
ref = f()
for rep in range(4):
    o = f()
    for key in ('image_raw', 'image_depth', 'weights_samples'):
        a, b = ref[key], o[key]
        ne = (a != b)
        n = int(ne.sum())
        if n == 0:
            print(rep, key, 'identical')
            continue
        d = (a - b).abs()
        pos = ne.reshape(V, -1, 256, 256).any(1).nonzero()
        rows = pos[:, 1].unique().tolist()
        print(rep, key, 'differs in', n, 'values; max abs', float(d.max()), 'max rel', float((d / (a.abs() + 1e-12))[ne].max()),
              '| views', pos[:, 0].unique().tolist(), '| rows', rows[:12], '...' if len(rows) > 12 else '',
              '| first pixels', pos[:6].tolist())
    if DEBUG:
        ne = (ref['image_raw'] != o['image_raw']).reshape(V, 3, -1).any(1)            # [V, M]
        for v, m in ne.nonzero().tolist()[:8]:
            for name, key in (('coarse sigma', 'coarse_densities'), ('fine depth', 'fine_depths')):
                a, b = ref['shape_synthesized'][key][v, m, :, 0], o['shape_synthesized'][key][v, m, :, 0]
                print(f'   ray ({v}, {m // 256}, {m % 256}) {name}: differs at {int((a != b).sum())} of 64 samples; finite {bool(torch.isfinite(a).all())}/{bool(torch.isfinite(b).all())};'
                      f' range [{float(a.min()):.5g}, {float(a.max()):.5g}] vs [{float(b.min()):.5g}, {float(b.max()):.5g}]'
                      + (f'; first differing samples {(a != b).nonzero().flatten().tolist()[:8]}' if bool((a != b).any()) else ''))
