#!/bin/bash
# same-box A/B of the ray-marcher: build/libln3d_prerank.so (r3 rank counting by full comparison, libm softplus / exp) vs the in-tree library;
# then the render tests against build/libln3d_slp.so (render.hip compiled WITHOUT -fno-slp-vectorize) for the determinism question
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r4_render_ab.log
: > $L
python -m pytest tests/test_render_gpu.py tests/test_geometry_gpu.py -q -x -k "render or 512" 2>&1 | tail -2 >> $L
for r in 1 2; do
  echo "== round $r: before" >> $L
  LN3D_LIB=build/libln3d_prerank.so python tools/render_bench.py 2>/dev/null >> $L
  echo "== round $r: after (in-tree)" >> $L
  python tools/render_bench.py 2>/dev/null >> $L
done
if [ -f build/libln3d_slp.so ]; then
  echo "== SLP build of render.hip: tests x3, then timing" >> $L
  cp ln3diff_amd/libln3d_hip.so /tmp/keep.so
  cp build/libln3d_slp.so ln3diff_amd/libln3d_hip.so
  for i in 1 2 3; do python -m pytest tests/test_render_gpu.py -q 2>&1 | tail -2 >> $L; done
  python tools/render_bench.py 256 2>/dev/null >> $L
  cp /tmp/keep.so ln3diff_amd/libln3d_hip.so
fi
cat $L
