#!/bin/bash
# r6: one bounded attempt at the lanes-48-63 irreproducibility of the SLP-vectorised render_kernel.
#  slp      = render.hip compiled with hipcc's SLP vectoriser ON (the failing build)
#  slppad   = the same + 88 KB of LDS padding per workgroup (ONE workgroup per CU)
#  slpocc1  = the same with __launch_bounds__(256, 1) (another register allocation)
#  nopad_pad= the shipped flags + the padding (control)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_slp.log; : > $L
for v in slp slppad slpocc1 nopad_pad; do
  for rep in 1 2; do
    echo "=== $v (process $rep)" >> $L
    LN3D_LIB=build/libln3d_$v.so timeout 300 python tools/render_repeat_diff.py debug 2>&1 | grep -v "identical" | head -30 >> $L
  done
done
echo "=== shipped" >> $L
timeout 300 python tools/render_repeat_diff.py 2>&1 | grep -c identical >> $L
cat $L
