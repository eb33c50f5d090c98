#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r4_gemm_ab5.log
: > $L
for r in 1 2; do echo "=== round $r p1" >> $L; timeout 200 build/gemm_bench_p1 3 GELU >> $L 2>&1;  echo "=== round $r p2" >> $L; timeout 200 build/gemm_bench_p2 3 GELU >> $L 2>&1; done
cat $L
