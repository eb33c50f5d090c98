#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_p4g.log; : > $L
timeout 300 build/gemm_bench_p4 2 "x18" >> $L 2>&1
for r in 1 2; do
for c in "fc1 GELU x16" "fc1 GELU x18" "i23d fc1 GELU M49152 x18" "i23d fc1 GELU M49152 x16"; do
timeout 300 build/gemm_bench_p4 3 "$c" 3000 >> $L 2>&1
done; done
grep -v LN3D $L
