#!/bin/bash
# r5 GPU call 6: bring-up of attn_kres1w_kernel (self-checking bench, both variants), the CFG-twins block-0 dedup test + same-box A/B
# (record of a GPU call: the temporary switch LN3D_NO_TWINS existed only for this measurement and has been removed from the product since)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 build/attn_bench > gpurun_out/r5_attn1w_bench.log 2>&1; echo "attn_bench rc $?" >> gpurun_out/r5_attn1w_bench.log
cat gpurun_out/r5_attn1w_bench.log
timeout 600 python -m pytest tests/test_dit_gpu.py tests/test_seams_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r5_dedup_tests.log
ROUNDS=3 timeout 900 bash tools/r4_ab_pipeline.sh nodedup:LN3D_NO_TWINS=1 dedup:LN3D_LANES=1 2>&1 | tail -12 | tee gpurun_out/r5_dedup_ab.log
