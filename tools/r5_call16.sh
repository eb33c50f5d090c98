#!/bin/bash
# r5 GPU call 16: ring configuration 13 (256x256, one wave per SIMD) in the product: tile-sweep tests, the I23D goldens, same-box A/B of
# (record of a GPU call: the temporary switch LN3D_NO_X13 existed only for this measurement and has been removed from the product since)
# the configs[2] bench line with / without it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "x13 or auto" 2>&1 | tail -4 | tee gpurun_out/r5_x13_tests.log
timeout 600 python -m pytest tests/test_i23d_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee -a gpurun_out/r5_x13_tests.log
ROUNDS=2 BENCH_ARGS="--workload i23d" timeout 900 bash tools/r4_ab_pipeline.sh base:LN3D_NO_X13=1 x13:LN3D_LANES=1 2>&1 | tail -6 | tee gpurun_out/r5_x13_ab.log
