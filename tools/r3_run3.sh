#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r3_attn3.log
for v in 6 7; do ATTN_BENCH_CASES=5 ATTN_BENCH_VAR=$v timeout 120 build/attn_bench >> gpurun_out/r3_attn3.log 2>&1; echo "var $v rc $?" >> gpurun_out/r3_attn3.log; done
for a in 1 2 4 8 3 12; do echo "== kres2 abl $a" >> gpurun_out/r3_attn3.log; ATTN_BENCH_CASES=1 ATTN_BENCH_VAR=6 timeout 60 build/attn_bench2_a$a >> gpurun_out/r3_attn3.log 2>&1; done
cat gpurun_out/r3_attn3.log
