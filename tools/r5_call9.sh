#!/bin/bash
# r5 GPU call 9: attn_kres1w_kernel option A/B (LN3D_K1W_OPT bits), self-checking, kres shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r5_attn1w_opt.log; : > $L
for rep in 1 2; do for o in 0 1 2 3 4 7; do
  echo "== OPT $o (rep $rep)" >> $L
  ATTN_BENCH_CASES=2 ATTN_BENCH_VAR=1 timeout 60 build/attn1w_o$o 2>&1 | grep "kres1w\|FAILED" >> $L
done; done
cat $L
