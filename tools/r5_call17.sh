#!/bin/bash
# r5 GPU call 17: PMC traffic of the GEMM entries on the final gemm_bf16.hip, the big-shape parity tests on the final tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/r5_pmc_gemm.sh 2>&1 | tail -5
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_geometry_gpu.py tests/test_fullsize_gpu.py tests/test_dit_gpu.py -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r5_x13_tests2.log
