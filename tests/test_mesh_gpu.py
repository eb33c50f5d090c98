"""Iso-surface extraction.  Classic marching cubes against golden meshes produced by scikit-image's classic (Lorensen & Cline)
implementation in the build container (tools/gen_mc_table.py: a smooth random field with ambiguous cells, an off-centre sphere):
the same triangles, in the same orientation, to 1e-4 of a cell.  Both extractors: analytic checks on a sphere - closed manifold
(every edge shared by exactly two triangles, Euler characteristic 2), outward orientation, vertices on the iso-level, area."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _canon(tri):
    """[T,3,3] triangle soup -> rotation-normalised (smallest vertex first, orientation kept), lexicographically sorted rows."""
    import numpy as np
    t = np.round(np.asarray(tri, dtype=np.float64), 4)
    out = np.empty_like(t)
    for i, tr in enumerate(t):
        k = min(range(3), key=lambda j: tuple(tr[j]))
        out[i] = np.roll(tr, -k, axis=0)
    flat = out.reshape(len(out), 9)
    return flat[np.lexsort(flat.T[::-1])]


@pytest.mark.parametrize("name", ["field", "sphere"])
def test_marching_cubes_vs_classic_golden(hip_lib, name):
    import numpy as np
    from conftest import golden
    from ln3diff_amd.mesh import extract_isosurface
    g = golden('mcubes_classic')
    sigma = torch.from_numpy(g[name + '_sigma']).cuda()
    v, f = extract_isosurface(sigma, float(g[name + '_level']), method='cubes')
    tri = v[f].cpu().numpy()
    ref = g[name + '_tri']
    assert tri.shape == ref.shape, (tri.shape, ref.shape)
    assert v.shape[0] == int(g[name + '_nverts'])                      # welded by grid edge: the same vertex count
    a, b = _canon(tri), _canon(ref)
    assert np.abs(a - b).max() < 2e-4, np.abs(a - b).max()            # same triangles, same winding


@pytest.mark.parametrize("method", ["tetra", "cubes"])
def test_sphere_isosurface(hip_lib, method):
    from ln3diff_amd.mesh import extract_isosurface
    G, r = 48, 15.3
    ax = torch.arange(G, dtype=torch.float32) - (G - 1) / 2 + 0.123
    X, Y, Z = torch.meshgrid(ax, ax + 0.05, ax - 0.07, indexing='ij')
    d = torch.sqrt(X * X + Y * Y + Z * Z)
    sigma = (10.0 + (r - d)).cuda()                  # > 10 inside the sphere
    v, f = extract_isosurface(sigma, 10.0, method=method)
    v, f = v.cpu(), f.cpu()
    assert f.shape[0] > (1000 if method == 'tetra' else 400)
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
