#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/r3_attn5.log
for a in 2 6 10 14 18 30 31 95; do echo "== kres3 abl $a" >> gpurun_out/r3_attn5.log; ATTN_BENCH_CASES=1 ATTN_BENCH_VAR=8 timeout 60 build/attn_bench3_a$a >> gpurun_out/r3_attn5.log 2>&1; done
grep "abl\|kres3" gpurun_out/r3_attn5.log | grep -v NONDET
