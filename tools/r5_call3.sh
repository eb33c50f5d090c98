#!/bin/bash
# r5 GPU call 3: SLP + fence repeatability, 100-scene sweep on the shipped build, residual-prefetch A/B (kernel level and whole line)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash tools/r5_slp_fence.sh > /dev/null 2>&1
timeout 600 python -m pytest tests/test_render_gpu.py -q -k "sweep_100" -s 2>&1 | tail -4 > gpurun_out/r5_c3_sweep.log
: > gpurun_out/r5_c3_gemm.log
for r in 1 2; do for b in gemm_bench_nopre gemm_bench; do echo "== $b round $r" >> gpurun_out/r5_c3_gemm.log; timeout 120 build/$b 5 "GATE_RES" 2>&1 | grep -v x14 >> gpurun_out/r5_c3_gemm.log; done; done
ROUNDS=3 timeout 900 bash tools/r4_ab_pipeline.sh nopre:LN3D_LIB=build/libln3d_nopre.so pre:LN3D_LANES=1 > /dev/null 2>&1
cp gpurun_out/r4_ab_pipeline3.log gpurun_out/r5_c3_pre_ab.log
cat gpurun_out/r5_slp_fence.log gpurun_out/r5_c3_sweep.log gpurun_out/r5_c3_gemm.log gpurun_out/r5_c3_pre_ab.log
