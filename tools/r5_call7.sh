#!/bin/bash
# r5 GPU call 7: bring-up of attn_kres1w_kernel - self-checking bench, shipped kernel vs the one-wave-per-SIMD kernel, kres shapes only
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ATTN_BENCH_CASES=5 timeout 200 build/attn_bench > gpurun_out/r5_attn1w_bench.log 2>&1; echo "attn_bench rc $?" >> gpurun_out/r5_attn1w_bench.log
cat gpurun_out/r5_attn1w_bench.log
