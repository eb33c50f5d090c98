#!/bin/bash
# r6, second (last) GPU call on the SLP irreproducibility: is it the LDS padding itself (an out-of-bounds LDS access absorbed by it) or
# the occupancy it forces?  slppad4k = SLP build + 4 KB of padding (still several workgroups per CU); slpw8pad = SLP build, 8 waves per
# workgroup + 80 KB of padding (ONE workgroup per CU but TWO waves per SIMD).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_slp2.log; : > $L
for v in slppad4k slpw8pad slppad; do
  for rep in 1 2; do
    echo "=== $v (process $rep)" >> $L
    LN3D_LIB=build/libln3d_$v.so timeout 300 python tools/render_repeat_diff.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-160 | head -14 >> $L
  done
done
cat $L
