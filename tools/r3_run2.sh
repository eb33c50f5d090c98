#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r3_abl2.log
for o in 0 1 2; do for a in 0 1 2 4 8 16 32 3 12 28; do
  echo "== order $o abl $a" >> gpurun_out/r3_abl2.log
  ATTN_BENCH_CASES=1 ATTN_BENCH_VAR=3 timeout 60 build/attn_bench_o${o}_a${a} >> gpurun_out/r3_abl2.log 2>&1
done; done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/r3_pytest_k2.log 2>&1; echo "pytest kernels rc $?"
tail -3 gpurun_out/r3_pytest_k2.log
grep -A2 "== order" gpurun_out/r3_abl2.log | grep "order\|kres" | paste - - | awk '{print $3, $5, $0}' | cut -c1-200
