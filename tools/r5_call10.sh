#!/bin/bash
# r5 GPU call 10: attn_kres1w_kernel - spread fragment reads (LN3D_K1W_OPT & 8) vs blocks of four, and which reads cost what
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r5_attn1w_opt2.log; : > $L
for rep in 1 2; do for b in o0 o8 o10 a256 a512 a8 o8a8; do
  echo "== $b (rep $rep)" >> $L
  ATTN_BENCH_CASES=2 ATTN_BENCH_VAR=1 timeout 60 build/attn1w_$b 2>&1 | grep "kres1w\|FAILED" >> $L
done; done
cat $L
