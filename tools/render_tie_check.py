#!/usr/bin/env python
"""Rays with bit-equal fine depths (the merge's tie path) against the CPU oracle, for the build LN3D_LIB selects (tests/test_render_gpu.py has
the same scene as a test of the in-tree build).  The oracle is the checker here, as in the tests."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LN3D_LIB'):
    from ln3diff_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['LN3D_LIB'])
from oracle import render as orender                                              # noqa: E402
from ln3diff_amd.nsr.triplane import Triplane, draw_render_noise                  # noqa: E402
from ln3diff_amd.synth import synth_input, orbit_cameras, synth_state_dict       # noqa: E402

res, V = 24, 2
sd = synth_state_dict({'net.0.weight': (64, 32), 'net.0.bias': (64,), 'net.2.weight': (4, 64), 'net.2.bias': (4,)}, 0)
sd['net.2.bias'] = sd['net.2.bias'].clone(); sd['net.2.bias'][0] += 4.0
tp = Triplane(img_resolution=res); tp.decoder.load_state_dict(sd); tp = tp.cuda()
planes = synth_input('planes', (V, 96, 128, 128), 3, 4.0)
cams = orbit_cameras(V)
for ties in (False, True):
    gen = torch.Generator().manual_seed(11)
    j, u = draw_render_noise(V, res * res, 64, generator=gen)
    u = u.clone()
    if ties:
        u[:, 7] = u[:, 3]; u[::2, 40] = u[::2, 3]
    ref = orender.triplane_render(planes, {k: v.float() for k, v in sd.items()}, cams, res, j.unsqueeze(-1), u)
    a = tp(planes.cuda(), cams.cuda(), jitter=j, u_fine=u)
    for key in ('image_raw', 'image_depth', 'weights_samples'):
        e = float((a[key].cpu().double() - ref[key].double()).norm() / ref[key].double().norm())
        print(f'ties {ties!s:5s} {key:16s} rel-L2 vs oracle {e:.3e}')
