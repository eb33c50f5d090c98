#!/bin/bash
# r6: ray head - uniform wave index + all per-ray loads in one batch (in-tree) against the previous build (ab/libln3d_r6b.so): hashes, tests, timing x2
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_head.log; : > $L
echo "== hashes: r6b then in-tree" >> $L
LN3D_LIB=ab/libln3d_r6b.so timeout 300 python tools/render_hash.py 2>&1 | grep -v amdgpu.ids >> $L
timeout 300 python tools/render_hash.py 2>&1 | grep -v amdgpu.ids >> $L
echo "== tests" >> $L
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_geometry_gpu.py tests/test_seams_gpu.py -q -x 2>&1 | tail -4 >> $L
for r in 1 2; do
  echo "== round $r: r6b" >> $L
  LN3D_LIB=ab/libln3d_r6b.so timeout 300 python tools/render_bench.py 2>/dev/null >> $L
  echo "== round $r: in-tree" >> $L
  timeout 300 python tools/render_bench.py 2>/dev/null >> $L
done
cat $L
