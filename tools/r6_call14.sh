#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_p4_other_shapes.log; : > $L
for r in 1 2; do for c in "xl2 fc1 GELU 12288x4608x1152" "dit2 fc1 GELU 24576x4096x1024"; do timeout 300 build/gemm_bench_p4 3 "$c" 2000 >> $L 2>&1; done; done
grep -v LN3D $L
