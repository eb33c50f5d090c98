#!/bin/bash
# ray-marcher occupancy variants (build/libln3d_w<WPB>o<OCC>.so: -DRENDER_WPB waves per workgroup sharing one decoder image,
# -DRENDER_OCC waves per SIMD asked of hipcc) against the in-tree library, same box; tests first
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r4_render_occ.log
: > $L
cp ln3diff_amd/libln3d_hip.so /tmp/keep.so
for v in "$@"; do
  cp build/libln3d_$v.so ln3diff_amd/libln3d_hip.so
  echo "== $v: $(python -m pytest tests/test_render_gpu.py -q 2>&1 | tail -1)" >> $L
done
cp /tmp/keep.so ln3diff_amd/libln3d_hip.so
for r in 1 2; do
  echo "== round $r in-tree" >> $L; python tools/render_bench.py 2>/dev/null >> $L
  for v in "$@"; do echo "== round $r $v" >> $L; LN3D_LIB=build/libln3d_$v.so python tools/render_bench.py 2>/dev/null >> $L; done
done
cat $L
