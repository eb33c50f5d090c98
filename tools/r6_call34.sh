#!/bin/bash
# r6: ablation builds of the r6 ray-marcher (wrong images by construction): 1 = no decoder MLP, 2 = no texel loads, 3 = neither, 4 = no compositing;
# seq = decoder of point tile 0 behind the whole gather instead of under its second half
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_abl.log; : > $L
for r in 1 2; do
for v in "" ab/libln3d_seq.so ab/libln3d_abl1.so ab/libln3d_abl2.so ab/libln3d_abl3.so ab/libln3d_abl4.so; do
  echo "== ${v:-in-tree}" >> $L
  LN3D_LIB=$v timeout 300 python tools/render_bench.py 256 2>/dev/null >> $L
done
done
cat $L
