#!/bin/bash
# r6 bisect of the in-tree ray-marcher's launch-to-launch differences: A = every tap weight in its own register (no packed-fp32 source taken
# from the HIGH half of a register pair through op_sel), B = texel loads through 64-bit vector addresses, C = compare / select softplus
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_det2.log; : > $L
for v in ab/libln3d_varA.so ab/libln3d_varB.so ab/libln3d_varC.so; do
  echo "== ${v:-in-tree}" >> $L
  LN3D_LIB=$v timeout 300 python tools/render_repeat_diff.py 2>&1 | grep -v "^$" | cut -c1-200 >> $L
done
cat $L
