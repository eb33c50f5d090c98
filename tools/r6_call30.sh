#!/bin/bash
# r6: the one-flag reproducer of profiles/r6_render_opsel.md on the final tree (repro build, repro build at one wave per SIMD, shipped build)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_opsel_repro.log; : > $L
for v in ab/libln3d_opsel_repro.so ab/libln3d_opsel_repro_1w.so ""; do
  echo "== ${v:-in-tree}" >> $L
  LN3D_LIB=$v timeout 300 python tools/render_repeat_diff.py 2>&1 | grep -v "^$\|amdgpu.ids" | cut -c1-160 >> $L
done
echo "== new tie test + render tests, in-tree" >> $L
timeout 900 python -m pytest tests/test_render_gpu.py -q 2>&1 | tail -3 >> $L
cat $L
