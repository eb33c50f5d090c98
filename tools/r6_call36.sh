#!/bin/bash
# r6: the shipped ray-marcher = sequential decoder, 8-wave workgroups, 4 waves per SIMD: tests, hashes, timing (r5 render.hip for reference), generic presets
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_final.log; : > $L
timeout 2400 python -m pytest tests/test_render_gpu.py tests/test_geometry_gpu.py tests/test_seams_gpu.py tests/test_unet_gpu.py tests/test_entry_gpu.py tests/test_mesh_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -4 >> $L
echo "== hashes in-tree" >> $L
timeout 300 python tools/render_hash.py 2>&1 | grep -v amdgpu.ids >> $L
echo "== repeat-diff in-tree" >> $L
timeout 300 python tools/render_repeat_diff.py 2>&1 | grep -v "^$\|amdgpu.ids" | cut -c1-120 >> $L
for r in 1 2; do
  echo "== round $r: r5 render.hip" >> $L
  LN3D_LIB=ab/libln3d_render_r5.so timeout 300 python tools/render_bench.py 2>/dev/null >> $L
  echo "== round $r: in-tree" >> $L
  timeout 300 python tools/render_bench.py 2>/dev/null >> $L
done
for p in objv128 shapenet; do RENDER_PRESET=$p timeout 300 python tools/render_bench.py 256 2>/dev/null >> $L; done
cat $L
