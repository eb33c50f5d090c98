#!/bin/bash
# r6 ray-marcher: setup hand-off through the feature tile + scalar-base loads + decoder of point tile 0 under the loads of tile 1 + med3 softplus
# (in-tree) against ab/libln3d_render_r5.so (= HEAD's render.hip in the same library): render / geometry tests, then same-box timing x2
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_ab.log; : > $L
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_geometry_gpu.py tests/test_seams_gpu.py -q -x 2>&1 | tail -5 >> $L
for r in 1 2; do
  echo "== round $r: r5 render (ab/libln3d_render_r5.so)" >> $L
  LN3D_LIB=ab/libln3d_render_r5.so timeout 300 python tools/render_bench.py 2>/dev/null >> $L
  echo "== round $r: in-tree" >> $L
  timeout 300 python tools/render_bench.py 2>/dev/null >> $L
done
echo "== generic kernel presets, r5 then in-tree" >> $L
for p in objv128 shapenet; do
  RENDER_PRESET=$p LN3D_LIB=ab/libln3d_render_r5.so timeout 300 python tools/render_bench.py 256 2>/dev/null >> $L
  RENDER_PRESET=$p timeout 300 python tools/render_bench.py 256 2>/dev/null >> $L
done
cat $L
