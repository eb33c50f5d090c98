#!/bin/bash
# r6: the sequential-decoder build of the ray-marcher fits 128 VGPRs with 6 spills: 8-wave workgroups at 4 waves per SIMD (r4 measured that
# occupancy 8 % slower at 28 spills) against the shipped 4-wave / 3-per-SIMD build and the sequential build at 3 per SIMD
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_occ4.log; : > $L
for r in 1 2 3; do
for v in "" ab/libln3d_seq.so ab/libln3d_seq_w8o4.so ab/libln3d_il_w8o4.so; do
  echo "== ${v:-in-tree}" >> $L
  LN3D_LIB=$v timeout 300 python tools/render_bench.py 2>/dev/null >> $L
done
done
echo "== hashes seq_w8o4" >> $L
LN3D_LIB=ab/libln3d_seq_w8o4.so timeout 300 python tools/render_hash.py 2>&1 | grep -v amdgpu.ids >> $L
cat $L
