#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_p4_v1.log; : > $L
for r in 1 2; do
timeout 300 build/gemm_bench_p4 3 "x16" >> $L 2>&1
timeout 100 build/gemm_bench_p4 3 "fc1 GELU_ERF" >> $L 2>&1
timeout 100 build/gemm_bench_p4 3 "fc1 plain" >> $L 2>&1
timeout 100 build/gemm_bench_p4 3 "i23d fc1 GELU M49152" >> $L 2>&1
done
cat $L
