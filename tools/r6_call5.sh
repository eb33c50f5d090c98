#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_p4_timeline.log; : > $L
for c in "fc1 GELU x16" "fc1 plain x16" "i23d fc1 GELU M49152 x16"; do
timeout 300 build/gemm_bench_p4abl8 2 "$c" 2000 >> $L 2>&1
timeout 300 build/gemm_bench_p4abl1 2 "$c" 2000 >> $L 2>&1
done
grep -v LN3D $L
