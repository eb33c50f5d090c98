#!/bin/bash
# r6: merge ranks by histogram + strict count with a rank-total check (in-tree) against the previous r6 build (ab/libln3d_r6a.so): output hashes
# (incl. rays with bit-equal fine depths), render tests, timing x2
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_rank.log; : > $L
echo "== hashes: r6a then in-tree" >> $L
LN3D_LIB=ab/libln3d_r6a.so timeout 300 python tools/render_hash.py 2>&1 | grep -v amdgpu.ids >> $L
timeout 300 python tools/render_hash.py 2>&1 | grep -v amdgpu.ids >> $L
echo "== tests" >> $L
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_geometry_gpu.py tests/test_seams_gpu.py -q -x 2>&1 | tail -4 >> $L
for r in 1 2; do
  echo "== round $r: r6a" >> $L
  LN3D_LIB=ab/libln3d_r6a.so timeout 300 python tools/render_bench.py 2>/dev/null >> $L
  echo "== round $r: in-tree" >> $L
  timeout 300 python tools/render_bench.py 2>/dev/null >> $L
done
cat $L
