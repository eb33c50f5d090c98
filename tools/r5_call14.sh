#!/bin/bash
# r5 GPU call 14: the 256x256 / 4-wave (one wave per SIMD, 128x128 per wave, AGPR accumulators) ring configuration x13 against the
# shipped tiles, isolated (self-checking gemm_bench), + the vendor-library reference point
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 build/gemm_bench 3 > gpurun_out/r5_gemm_x13.log 2>&1; echo "rc $?" >> gpurun_out/r5_gemm_x13.log
cat gpurun_out/r5_gemm_x13.log | cut -c1-200
