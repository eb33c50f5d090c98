#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_hash.log; : > $L
for v in ab/libln3d_render_r5.so ab/libln3d_r6a.so ""; do
  echo "== ${v:-in-tree}" >> $L
  LN3D_LIB=$v timeout 300 python tools/render_hash.py 2>&1 | grep -v amdgpu.ids >> $L
done
cat $L
