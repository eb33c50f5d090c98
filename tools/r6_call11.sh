#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_p4g_njd.log; : > $L
timeout 300 build/gemm_bench_p4 2 "x18" >> $L 2>&1
timeout 300 build/gemm_bench_p4j1 2 "x18" >> $L 2>&1
for r in 1 2; do
echo "--- NJD=2" >> $L; timeout 300 build/gemm_bench_p4 3 "fc1 GELU x18" 3000 >> $L 2>&1
echo "--- NJD=1" >> $L; timeout 300 build/gemm_bench_p4j1 3 "fc1 GELU x18" 3000 >> $L 2>&1
echo "--- x16" >> $L; timeout 300 build/gemm_bench_p4 3 "fc1 GELU x16" 3000 >> $L 2>&1
echo "--- x17" >> $L; timeout 300 build/gemm_bench_p4 3 "fc1 GELU x17" 3000 >> $L 2>&1
done
grep -v LN3D $L
