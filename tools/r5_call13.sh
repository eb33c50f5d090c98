#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/gemm_vs_blas.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r5_gemm_vs_blas.log
