#!/bin/bash
# r6: the standalone op_sel probe; then the fixed in-tree ray-marcher (odd-register weights copied to even registers): repeat-diff, render /
# geometry / seam / full-size tests, same-box timing against ab/libln3d_render_r5.so
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_fix.log; : > $L
echo "== tools/pk_opsel_hi_probe.hip" >> $L
timeout 300 ab/pk_opsel_hi_probe >> $L 2>&1
echo "== in-tree: repeat-diff" >> $L
timeout 300 python tools/render_repeat_diff.py 2>&1 | grep -v "^$" | cut -c1-200 >> $L
echo "== tests" >> $L
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_geometry_gpu.py tests/test_seams_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -5 >> $L
for r in 1 2; do
  echo "== round $r: r5 render (ab/libln3d_render_r5.so)" >> $L
  LN3D_LIB=ab/libln3d_render_r5.so timeout 300 python tools/render_bench.py 2>/dev/null >> $L
  echo "== round $r: in-tree" >> $L
  timeout 300 python tools/render_bench.py 2>/dev/null >> $L
done
cat $L
