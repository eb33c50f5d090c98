#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 200 build/gemm_bench 3 x15; timeout 200 build/gemm_bench 3 "fc2 GATE_RES"; timeout 100 build/gemm_bench 3 "i23d fc2") > gpurun_out/r5_gemm_x15.log 2>&1
cat gpurun_out/r5_gemm_x15.log | cut -c1-200
