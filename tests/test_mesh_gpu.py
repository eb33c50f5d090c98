"""Iso-surface extraction (marching tetrahedra): analytic checks on a sphere - closed manifold (every edge shared by
exactly two triangles, Euler characteristic 2), outward orientation, vertices on the iso-level, area ~ 4 pi r^2."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sphere_isosurface(hip_lib):
    from ln3diff_amd.mesh import extract_isosurface
    G, r = 48, 15.3
    ax = torch.arange(G, dtype=torch.float32) - (G - 1) / 2 + 0.123
    X, Y, Z = torch.meshgrid(ax, ax + 0.05, ax - 0.07, indexing='ij')
    d = torch.sqrt(X * X + Y * Y + Z * Z)
    sigma = (10.0 + (r - d)).cuda()                  # > 10 inside the sphere
    v, f = extract_isosurface(sigma, 10.0)
    v, f = v.cpu(), f.cpu()
    assert f.shape[0] > 1000
    # vertices lie on the iso-level of the (trilinear) field: distance to centre ~ r
    c = torch.tensor([(G - 1) / 2 - 0.123, (G - 1) / 2 - 0.123 - 0.05, (G - 1) / 2 - 0.123 + 0.07])
    rad = (v - c).norm(dim=1)
    assert float((rad - r).abs().max()) < 0.05
    # closed 2-manifold: each undirected edge appears exactly twice, once per direction
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = e.min(1).values * v.shape[0] + e.max(1).values
    _, cnt = torch.unique(key, return_counts=True)
    assert int(cnt.min()) == 2 and int(cnt.max()) == 2
    dkey = e[:, 0] * v.shape[0] + e[:, 1]
    assert torch.unique(dkey).numel() == dkey.numel()          # consistent orientation
    assert v.shape[0] - e.shape[0] // 2 + f.shape[0] == 2      # Euler characteristic of a sphere
    # outward normals and area
    p0, p1, p2 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = torch.linalg.cross(p1 - p0, p2 - p0)
    assert bool(((n * ((p0 + p1 + p2) / 3 - c)).sum(1) > 0).all())
    area = float(0.5 * n.norm(dim=1).sum())
    assert abs(area / (4 * math.pi * r * r) - 1) < 0.02


def test_export_obj(hip_lib, tmp_path):
    from ln3diff_amd.mesh import export_mesh
    from ln3diff_amd.synth import synth_input
    from test_decode_gpu import build_decoder
    dec = build_decoder(128, 2, 2).cuda()
    dec.triplane_decoder.decoder.net[2].bias.data[0] += 10.0
    pcl = (synth_input('pcl', (1, 3, 128, 128, 32), 9, 4.0)).cuda()
    nv, nf = export_mesh(dec, {'planes_channel_last': pcl}, str(tmp_path / 'm.obj'), grid_size=32, thr=10.0)
    txt = open(tmp_path / 'm.obj').read().splitlines()
    assert nv > 0 and nf > 0 and sum(l.startswith('v ') for l in txt) == nv and sum(l.startswith('f ') for l in txt) == nf
