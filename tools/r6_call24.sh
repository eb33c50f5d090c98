#!/bin/bash
# r6: general attention kernel with the cheaper softmax (scale folded into Q, -m as the C operand, v_max3, packed row sums) against HEAD's
# attention.hip (ab/libln3d_attn_r5.so); the extended op_sel probe; attention tests
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_attn_general.log; : > $L
echo "== tools/pk_opsel_hi_probe.hip (extended)" >> $L
timeout 300 ab/pk_opsel_hi_probe 2>&1 | tail -8 >> $L
for r in 1 2; do
  echo "== round $r: r5 attention.hip" >> $L
  LN3D_LIB=ab/libln3d_attn_r5.so timeout 300 python tools/attn_general_ab.py 2>&1 | grep -v amdgpu.ids >> $L
  echo "== round $r: in-tree" >> $L
  timeout 300 python tools/attn_general_ab.py 2>&1 | grep -v amdgpu.ids >> $L
done
echo "== tests" >> $L
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "attn or attention" 2>&1 | tail -4 >> $L
cat $L
