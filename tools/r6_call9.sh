#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_p4_nj3.log; : > $L
timeout 300 build/gemm_bench_p4 2 "x17" >> $L 2>&1
timeout 300 build/gemm_bench_p4 2 "small" >> $L 2>&1
for r in 1 2; do
for c in "fc1 GELU x16" "fc1 GELU x17" "fc1 plain x16" "fc1 plain x17" "qkv plain x17" "qkv plain x12" "qkv HEADS x12"; do
timeout 300 build/gemm_bench_p4 3 "$c" 3000 >> $L 2>&1
done; done
grep -v LN3D $L
