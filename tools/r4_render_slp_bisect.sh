#!/bin/bash
# determinism of the ray-marcher for builds of render.hip with hipcc's SLP vectoriser limited by tree cost (build/libln3d_slp_t<T>.so:
# -mllvm -slp-threshold=T keeps only trees that gain more than T) - profiles/r4_render_spill.md
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r4_render_slp_bisect.log
: > $L
cp ln3diff_amd/libln3d_hip.so /tmp/keep.so
for v in "$@"; do
  cp build/libln3d_slp_$v.so ln3diff_amd/libln3d_hip.so
  for i in 1 2; do
    r=$(python -m pytest tests/test_render_gpu.py -q -k 256_properties 2>&1 | tail -1)
    echo "$v run $i: $r" >> $L
  done
done
cp /tmp/keep.so ln3diff_amd/libln3d_hip.so
cat $L
