// Iso-surface extraction on the sigma grid for gfx950 (the "mesh" step of the sampling drivers:
// nsr/train_util_diffusion.py:208-248 calls PyMCubes 0.1.4 `marching_cubes(sigma[G,G,G], thr)`, a third-party package
// absent from the reference tree: parity with PyMCubes itself is unpinned).  Two extractors share the count / emit passes:
//  * classic marching CUBES (Lorensen & Cline, the algorithm mcubes implements): 256-case triangle table mc_table.h, pinned
//    against scikit-image's classic implementation (tools/gen_mc_table.py, tests/golden/mcubes_classic.npz) - the default;
//  * marching TETRAHEDRA (round 1; no ambiguous cases, watertight by construction) on the Kuhn decomposition of each cell
// (6 tetrahedra around the 0-7 diagonal; face-consistent across cells, no 256-case table): every tetrahedron emits 0, 1
// or 2 triangles; vertices are identified by the grid edge they lie on (key = min_vertex_id * G^3 + max_vertex_id), so the
// host welds them with one unique() and no floating-point comparison.  Two passes (count, emit) around a prefix sum.
#include "common.h"
#include "../../include/ln3d.h"
#include "mc_table.h"

__constant__ int kTet[6][4] = {{0, 1, 3, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 6, 7}, {0, 4, 5, 7}, {0, 1, 5, 7}};

struct MeshP { const float* sigma; int G; float thr; };

__device__ __forceinline__ void cell_corners(const MeshP& p, int64_t cell, float v[8], int64_t gid[8], int& cx, int& cy, int& cz) {
  const int Gc = p.G - 1;
  cz = (int)(cell % Gc); cy = (int)((cell / Gc) % Gc); cx = (int)(cell / ((int64_t)Gc * Gc));   // sigma[x][y][z] (indexing='ij')
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int x = cx + (c & 1), y = cy + ((c >> 1) & 1), z = cz + (c >> 2);
    gid[c] = ((int64_t)x * p.G + y) * p.G + z;
    v[c] = p.sigma[gid[c]];
  }
}

__global__ void mesh_count_kernel(MeshP p, int64_t ncell, int32_t* counts) {
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  float v[8]; int64_t gid[8]; int cx, cy, cz;
  cell_corners(p, cell, v, gid, cx, cy, cz);
  int n = 0;
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    int in = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) in += v[kTet[t][k]] > p.thr;
    n += (in == 1 || in == 3) ? 1 : (in == 2 ? 2 : 0);
  }
  counts[cell] = n;
}

__device__ __forceinline__ void edge_point(const MeshP& p, const float v[8], const int64_t gid[8], int cx, int cy, int cz, int a, int b,
                                           float out[3], int64_t& key) {
  if (gid[a] > gid[b]) { const int t = a; a = b; b = t; }            // canonical orientation: identical bits from every cell
  const float t = (p.thr - v[a]) / (v[b] - v[a]);
  const float ax = cx + (a & 1), ay = cy + ((a >> 1) & 1), az = cz + (a >> 2);
  const float bx = cx + (b & 1), by = cy + ((b >> 1) & 1), bz = cz + (b >> 2);
  out[0] = ax + t * (bx - ax); out[1] = ay + t * (by - ay); out[2] = az + t * (bz - az);
  key = gid[a] * ((int64_t)p.G * p.G * p.G) + gid[b];
}

__global__ void mesh_emit_kernel(MeshP p, int64_t ncell, const int64_t* offsets, float* tri_pos, int64_t* tri_key) {
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  float v[8]; int64_t gid[8]; int cx, cy, cz;
  cell_corners(p, cell, v, gid, cx, cy, cz);
  int64_t o = cell == 0 ? 0 : offsets[cell - 1];                      // offsets = inclusive prefix sum of counts
  for (int t = 0; t < 6; ++t) {
    int ins[4], outs[4], ni = 0, no = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = kTet[t][k];
      if (v[c] > p.thr) ins[ni++] = c; else outs[no++] = c;
    }
    if (ni == 0 || ni == 4) continue;
    float q[4][3]; int64_t kk[4]; int nq;
    if (ni == 1) { nq = 3; for (int j = 0; j < 3; ++j) edge_point(p, v, gid, cx, cy, cz, ins[0], outs[j], q[j], kk[j]); }
    else if (ni == 3) { nq = 3; for (int j = 0; j < 3; ++j) edge_point(p, v, gid, cx, cy, cz, outs[0], ins[j], q[j], kk[j]); }
    else {
      nq = 4;
      edge_point(p, v, gid, cx, cy, cz, ins[0], outs[0], q[0], kk[0]);
      edge_point(p, v, gid, cx, cy, cz, ins[0], outs[1], q[1], kk[1]);
      edge_point(p, v, gid, cx, cy, cz, ins[1], outs[1], q[2], kk[2]);
      edge_point(p, v, gid, cx, cy, cz, ins[1], outs[0], q[3], kk[3]);
    }
    // orient: normal must point from the inside (sigma > thr) towards the outside
    float ci[3] = {0, 0, 0}, co[3] = {0, 0, 0};
    for (int j = 0; j < ni; ++j) { ci[0] += (ins[j] & 1); ci[1] += ((ins[j] >> 1) & 1); ci[2] += (ins[j] >> 2); }
    for (int j = 0; j < no; ++j) { co[0] += (outs[j] & 1); co[1] += ((outs[j] >> 1) & 1); co[2] += (outs[j] >> 2); }
    const float dir[3] = {co[0] / no - ci[0] / ni, co[1] / no - ci[1] / ni, co[2] / no - ci[2] / ni};
    const int ntri = nq == 3 ? 1 : 2;
    for (int tr = 0; tr < ntri; ++tr) {
      int i0 = 0, i1 = tr == 0 ? 1 : 2, i2 = tr == 0 ? 2 : 3;
      const float e1[3] = {q[i1][0] - q[i0][0], q[i1][1] - q[i0][1], q[i1][2] - q[i0][2]};
      const float e2[3] = {q[i2][0] - q[i0][0], q[i2][1] - q[i0][1], q[i2][2] - q[i0][2]};
      const float nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
      if (nx * dir[0] + ny * dir[1] + nz * dir[2] < 0.f) { const int s = i1; i1 = i2; i2 = s; }
      const int idx[3] = {i0, i1, i2};
      for (int j = 0; j < 3; ++j) {
        tri_pos[(o * 3 + j) * 3 + 0] = q[idx[j]][0]; tri_pos[(o * 3 + j) * 3 + 1] = q[idx[j]][1]; tri_pos[(o * 3 + j) * 3 + 2] = q[idx[j]][2];
        tri_key[o * 3 + j] = kk[idx[j]];
      }
      ++o;
    }
  }
}

// ------------------------------------------------------------------ classic marching cubes
__device__ __forceinline__ int mc_case(const MeshP& p, const float v[8]) {
  int cs = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) cs |= (v[c] > p.thr) << c;
  return cs;
}
__global__ void mcubes_count_kernel(MeshP p, int64_t ncell, int32_t* counts) {
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  float v[8]; int64_t gid[8]; int cx, cy, cz;
  cell_corners(p, cell, v, gid, cx, cy, cz);
  counts[cell] = kMcCount[mc_case(p, v)];
}
__global__ void mcubes_emit_kernel(MeshP p, int64_t ncell, const int64_t* offsets, float* tri_pos, int64_t* tri_key) {
  const int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncell) return;
  float v[8]; int64_t gid[8]; int cx, cy, cz;
  cell_corners(p, cell, v, gid, cx, cy, cz);
  const int cs = mc_case(p, v);
  const int nt = kMcCount[cs];
  int64_t o = cell == 0 ? 0 : offsets[cell - 1];
  for (int t = 0; t < nt; ++t, ++o)
    for (int j = 0; j < 3; ++j) {
      const int e = kMcTri[cs][3 * t + j];
      const int ax = e >> 2, b0 = e & 1, b1 = (e >> 1) & 1;
      // edge parallel to axis ax; its other two coordinates (in axis order) are (b0, b1)
      const int a = ax == 0 ? (b0 << 1) | (b1 << 2) : (ax == 1 ? b0 | (b1 << 2) : b0 | (b1 << 1));
      const int b = a | (1 << ax);
      float q[3]; int64_t key;
      edge_point(p, v, gid, cx, cy, cz, a, b, q, key);
      tri_pos[(o * 3 + j) * 3 + 0] = q[0]; tri_pos[(o * 3 + j) * 3 + 1] = q[1]; tri_pos[(o * 3 + j) * 3 + 2] = q[2];
      tri_key[o * 3 + j] = key;
    }
}

extern "C" int ln3d_mcubes_count(const float* sigma, int G, float thr, int32_t* counts, void* stream) {
  if (!sigma || !counts || G < 2) return LN3D_ERR_BAD_ARG;
  MeshP p{sigma, G, thr};
  const int64_t ncell = (int64_t)(G - 1) * (G - 1) * (G - 1);
  hipLaunchKernelGGL(mcubes_count_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, ncell, counts);
  return ln3d_check_launch();
}
extern "C" int ln3d_mcubes_emit(const float* sigma, int G, float thr, const int64_t* offsets, float* tri_pos, int64_t* tri_key, void* stream) {
  if (!sigma || !offsets || !tri_pos || !tri_key || G < 2) return LN3D_ERR_BAD_ARG;
  MeshP p{sigma, G, thr};
  const int64_t ncell = (int64_t)(G - 1) * (G - 1) * (G - 1);
  hipLaunchKernelGGL(mcubes_emit_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, ncell, offsets, tri_pos, tri_key);
  return ln3d_check_launch();
}

extern "C" int ln3d_mesh_count(const float* sigma, int G, float thr, int32_t* counts, void* stream) {
  if (!sigma || !counts || G < 2) return LN3D_ERR_BAD_ARG;
  MeshP p{sigma, G, thr};
  const int64_t ncell = (int64_t)(G - 1) * (G - 1) * (G - 1);
  hipLaunchKernelGGL(mesh_count_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, ncell, counts);
  return ln3d_check_launch();
}
extern "C" int ln3d_mesh_emit(const float* sigma, int G, float thr, const int64_t* offsets, float* tri_pos, int64_t* tri_key, void* stream) {
  if (!sigma || !offsets || !tri_pos || !tri_key || G < 2) return LN3D_ERR_BAD_ARG;
  MeshP p{sigma, G, thr};
  const int64_t ncell = (int64_t)(G - 1) * (G - 1) * (G - 1);
  hipLaunchKernelGGL(mesh_emit_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, ncell, offsets, tri_pos, tri_key);
  return ln3d_check_launch();
}
