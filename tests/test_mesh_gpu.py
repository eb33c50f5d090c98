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


def test_mesh_from_grid_scale_rotation_colours_obj(hip_lib, tmp_path):
    """nsr/train_util_diffusion.py:221-244 after the iso-surface: vertices (v / (G-1) * 2 - 1) * 0.45, colours = the decoder's rgb
    re-queried at those points, rotation -90 degrees about x ((x, y, z) -> (x, z, -y)), .obj with per-vertex colours."""
    import numpy as np
    from ln3diff_amd.mesh import extract_isosurface, mesh_from_grid
    from ln3diff_amd.synth import synth_input
    from test_decode_gpu import build_decoder
    dec = build_decoder(128, 2, 2).cuda()
    dec.triplane_decoder.decoder.net[2].bias.data[0] += 10.0
    pcl = synth_input('pcl', (1, 3, 128, 128, 32), 9, 4.0).cuda()
    G = 24
    grid = dec.triplane_decode_grid({'planes_channel_last': pcl}, G)
    sigma = grid['sigma'][0].reshape(G, G, G)
    v, f, col = mesh_from_grid(dec, {'planes_channel_last': pcl}, sigma, G, thr=10.0, path=str(tmp_path / 'm.obj'))
    gv, gf = extract_isosurface(sigma.contiguous(), 10.0)
    assert f.shape[0] > 50 and np.array_equal(f, gf.cpu().numpy())
    world = (gv / (G - 1) * 2 - 1) * 0.45
    w = world.cpu().numpy()
    assert np.allclose(v, np.stack([w[:, 0], w[:, 2], -w[:, 1]], 1), atol=1e-6)
    rgb = dec.forward_points(pcl, world[None])['rgb'][0]
    assert np.array_equal(col, (rgb.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy())
    # the surface vertices sit on the iso-level of the decoder's density (trilinear interpolation of a smooth field: close)
    sig_v = dec.forward_points(pcl, world[None])['sigma'][0, :, 0]
    assert float((sig_v - 10.0).abs().median()) < 1.0
    lines = open(tmp_path / 'm.obj').read().splitlines()
    vl = [l.split() for l in lines if l.startswith('v ')]
    fl = [l.split() for l in lines if l.startswith('f ')]
    assert len(vl) == v.shape[0] and len(fl) == f.shape[0] and len(vl[0]) == 7
    assert np.allclose(np.array(vl[5][1:4], dtype=np.float64), v[5], atol=1e-5)
    assert np.allclose(np.array(vl[5][4:7], dtype=np.float64), col[5] / 255.0, atol=1e-3)
    assert [int(i) - 1 for i in fl[3][1:4]] == list(f[3])
