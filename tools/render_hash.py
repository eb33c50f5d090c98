#!/usr/bin/env python
"""SHA-256 of the ray-marcher's outputs on a fixed scene set (same-box check that a re-scheduling of the kernel left every bit in place:
run with LN3D_LIB = the other build and compare the lines)."""
import hashlib
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('LN3D_LIB'):
    from ln3diff_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['LN3D_LIB'])
from ln3diff_amd.nsr.triplane import Triplane          # noqa: E402
from ln3diff_amd.synth import synth_input, orbit_cameras            # noqa: E402

torch.manual_seed(0)                                  # the decoder is random-initialised: the same weights in every process
tp = Triplane(img_resolution=256).cuda()
tp.decoder.net[2].bias.data[0] += 4.0
for seed, res, V, radius in ((3, 256, 4, None), (5, 128, 6, None), (7, 512, 1, None), (9, 192, 3, None)):
    planes = synth_input('planes', (1, 96, 128, 128), seed, 4.0).cuda()
    pcl = tp.to_channel_last(planes)
    cams = orbit_cameras(V).cuda()
    g = torch.Generator(device='cuda').manual_seed(seed)
    j = torch.rand(V, res * res, 64, device='cuda', generator=g)
    u = torch.rand(V * res * res, 64, device='cuda', generator=g)
    u[:, 7] = u[:, 3]                                   # bit-equal fine depths on every ray: the tie path of the merge
    idx = torch.zeros(V, dtype=torch.int32, device='cuda')
    o = tp(c=cams, planes_channel_last=pcl, plane_index=idx, neural_rendering_resolution=res, jitter=j, u_fine=u)
    h = hashlib.sha256()
    for key in ('image_raw', 'image_depth', 'weights_samples'):
        h.update(o[key].contiguous().cpu().numpy().tobytes())
    print(f'seed {seed} res {res} V {V}: {h.hexdigest()[:32]}  mean {float(o["image_raw"].mean()):+.6f}')
