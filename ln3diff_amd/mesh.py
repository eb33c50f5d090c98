"""Mesh extraction from the sigma grid (the export_mesh branch of render_video_given_triplane,
nsr/train_util_diffusion.py:208-248): iso-surface at sigma = 10 on the G^3 grid, vertices mapped to the +-0.45 box,
coloured by re-querying the tri-plane, rotated -90 degrees about x, written as .obj with per-vertex colours.
The surface is classic marching cubes (what the reference's `mcubes.marching_cubes` is; method='cubes', the default) or marching
tetrahedra (method='tetra'); both run on the GPU in two passes around a prefix sum and weld vertices by grid-edge id."""
import math

import numpy as np
import torch

from . import ops


@torch.no_grad()
def extract_isosurface(sigma, thr=10.0, method='cubes'):
    """sigma [G,G,G] f32 device -> (verts [Nv,3] in grid coordinates, faces [Nf,3] int64).  Faces keep the emission order
    (cells in x-major order, the case table's triangle order inside a cell)."""
    count, emit = {'cubes': (ops.mcubes_count, ops.mcubes_emit), 'tetra': (ops.mesh_count, ops.mesh_emit)}[method]
    G = sigma.shape[0]
    dev = sigma.device
    sigma = sigma.contiguous().float()
    ncell = (G - 1) ** 3
    counts = torch.empty(ncell, dtype=torch.int32, device=dev)
    count(sigma, G, thr, counts)
    offs = torch.cumsum(counts.long(), 0)
    ntri = int(offs[-1])
    if ntri == 0:
        return torch.zeros(0, 3, device=dev), torch.zeros(0, 3, dtype=torch.long, device=dev)
    pos = torch.empty(ntri * 3, 3, device=dev)
    key = torch.empty(ntri * 3, dtype=torch.int64, device=dev)
    emit(sigma, G, thr, offs, pos, key)
    uniq, inv = torch.unique(key, return_inverse=True)                # weld by grid-edge id
    verts = torch.empty(uniq.shape[0], 3, device=dev)
    verts[inv] = pos                                                   # identical bits for every copy of a vertex
    faces = inv.view(ntri, 3)
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return verts, faces[ok]


def rotation_matrix_x(deg):
    a = math.radians(deg)
    return np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]], dtype=np.float32)


def write_obj(path, v, f, c):
    """Wavefront .obj with per-vertex colours (what trimesh.Trimesh(vertex_colors=...).export(path, 'obj') writes)."""
    with open(path, 'w') as fh:
        for i in range(v.shape[0]):
            fh.write('v %.6f %.6f %.6f %.4f %.4f %.4f\n' % (v[i, 0], v[i, 1], v[i, 2], c[i, 0], c[i, 1], c[i, 2]))
        for t in f:
            fh.write('f %d %d %d\n' % (t[0] + 1, t[1] + 1, t[2] + 1))


@torch.no_grad()
def mesh_from_grid(decoder, dec_out, sigma, grid_size, thr=10.0, sample_index=0, path=None, method='cubes'):
    """nsr/train_util_diffusion.py:221-244: iso-surface of the sigma grid at `thr`, vertices mapped to the +-0.45 box, coloured
    by re-querying the tri-plane at the vertices (forward_points), rotated -90 degrees about x.
    Returns (verts [Nv,3] float32 numpy, faces [Nf,3] int64 numpy, colors [Nv,3] uint8 numpy); writes `path` when given."""
    sigma = sigma.reshape(grid_size, grid_size, grid_size)
    verts, faces = extract_isosurface(sigma, thr, method)
    vtx = (verts / (grid_size - 1) * 2 - 1) * 0.45                       # g-objaverse scale
    pcl = dec_out.get('planes_channel_last')
    if pcl is None:
        from .nsr.triplane import Triplane
        pcl = Triplane.to_channel_last(dec_out['latent_after_vit'])
    pcl = pcl[sample_index:sample_index + 1]
    col = decoder.forward_points(pcl, vtx[None])['rgb'][0] if vtx.shape[0] else vtx
    colors = (col.clamp(0, 1) * 255).to(torch.uint8).cpu().numpy()
    v = (rotation_matrix_x(-90) @ vtx.cpu().numpy().T).T.astype(np.float32)
    f = faces.cpu().numpy()
    if path:
        write_obj(path, v, f, colors.astype(np.float32) / 255.0)
    return v, f, colors


@torch.no_grad()
def export_mesh(decoder, dec_out, path, grid_size=192, thr=10.0, sample_index=0):
    """decoder: the VAE decoder module; dec_out: its vit_decode_postprocess dict.  Writes `path` (.obj)."""
    pcl = dec_out['planes_channel_last'][sample_index:sample_index + 1]
    grid = decoder.triplane_decode_grid({'planes_channel_last': pcl}, grid_size)
    v, f, _ = mesh_from_grid(decoder, {'planes_channel_last': pcl}, grid['sigma'][0], grid_size, thr, 0, path)
    return v.shape[0], f.shape[0]
