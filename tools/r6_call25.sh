#!/bin/bash
# r6: general attention kernel at the 128-wide shapes: static priority for waves 4-7 / a 4-deep ring / both, against the in-tree build
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_attn_general2.log; : > $L
for r in 1 2; do
  for v in "" ab/libln3d_attn_p.so ab/libln3d_attn_n4.so ab/libln3d_attn_pn4.so; do
    echo "== round $r: ${v:-in-tree}" >> $L
    LN3D_LIB=$v timeout 300 python tools/attn_general_ab.py 2>&1 | grep -v amdgpu.ids | head -2 >> $L
  done
done
cat $L
