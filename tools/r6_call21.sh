#!/bin/bash
# r6: the in-tree ray-marcher (LDS setup hand-off + decoder under the gather) fails the launch-to-launch bitwise check like the SLP builds did.
# Variants of render.hip only: in-tree; inb carried in a VGPR; SGPR spills to memory instead of VGPR lanes; gather and decoder in sequence;
# one wave per SIMD (88 KB LDS padding)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_det.log; : > $L
for v in "" ab/libln3d_inbv.so ab/libln3d_nosgprvgpr.so ab/libln3d_seq.so ab/libln3d_pad1w.so ab/libln3d_render_r5.so; do
  echo "== ${v:-in-tree}" >> $L
  LN3D_LIB=$v timeout 300 python tools/render_repeat_diff.py 2>&1 | grep -v "^$" | cut -c1-400 >> $L
done
echo "== in-tree, debug" >> $L
timeout 300 python tools/render_repeat_diff.py debug 2>&1 | cut -c1-600 >> $L
cat $L
