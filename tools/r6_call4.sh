#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_p4_sustained.log; : > $L
for r in 1 2; do
for c in "fc1 GELU_ERF" "fc1 GELU x16" "fc1 plain x7" "fc1 plain x16" "fc1 plain x13" "i23d fc1 GELU M49152"; do
timeout 300 build/gemm_bench_p4 3 "$c" 3000 >> $L 2>&1
done; done
grep -v LN3D $L
