#!/bin/bash
# r6: 80-wide head storage in the general attention kernel (DiT-XL/2, U-Net 80-wide heads) against 128-wide storage; attention / DiT tests
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_attn_dh80.log; : > $L
for r in 1 2; do
  echo "== round $r: in-tree" >> $L
  timeout 300 python tools/attn_general_ab.py 2>&1 | grep -v amdgpu.ids >> $L
done
echo "== r5 attention.hip" >> $L
LN3D_LIB=ab/libln3d_attn_r5.so timeout 300 python tools/attn_general_ab.py 2>&1 | grep -v amdgpu.ids >> $L
echo "== tests" >> $L
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_unet_gpu.py -q -x 2>&1 | tail -6 >> $L
cat $L
