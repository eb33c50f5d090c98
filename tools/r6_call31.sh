#!/bin/bash
# r6: generic ray-marcher with the merged planes aliased onto the feature tile (15 KB per wave: two workgroups per CU) against ab/libln3d_r6a.so
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_generic2.log; : > $L
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_seams_gpu.py tests/test_unet_gpu.py tests/test_entry_gpu.py -q -x 2>&1 | tail -4 >> $L
for r in 1 2; do
  for p in objv128 shapenet eg3d48; do
    echo "== round $r $p: r6a then in-tree" >> $L
    RENDER_PRESET=$p LN3D_LIB=ab/libln3d_r6a.so timeout 300 python tools/render_bench.py 256 2>/dev/null >> $L
    RENDER_PRESET=$p timeout 300 python tools/render_bench.py 256 2>/dev/null >> $L
  done
done
cat $L
