#!/bin/bash
# round-3 GPU call 1: new attention kernel (self-checking bench), kernel tests, GEMM micro-benchmarks
mkdir -p gpurun_out
for v in 0 1 2 3 4 5; do
  ATTN_BENCH_VAR=$v timeout 120 build/attn_bench > gpurun_out/r3_attn1_v$v.log 2>&1; echo "attn_bench var $v rc $?" >> gpurun_out/r3_run1.log
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/r3_pytest_k1.log 2>&1; echo "pytest kernels rc $?" >> gpurun_out/r3_run1.log
KBENCH_VENDOR=0 timeout 300 python tools/kbench.py > gpurun_out/r3_kbench1.log 2>&1; echo "kbench rc $?" >> gpurun_out/r3_run1.log
cat gpurun_out/r3_run1.log
tail -5 gpurun_out/r3_pytest_k1.log
