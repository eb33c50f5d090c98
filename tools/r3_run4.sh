#!/bin/bash
# PMC passes (separate counter sets, --kernel-trace only) over the attention bench: kres3 lockstep (var 8) and phased (var 9)
for v in 8 9; do PMC_TAG=r3_pmc_attn_v$v ATTN_BENCH_VAR=$v bash tools/pmc_attn.sh > gpurun_out/r3_pmc_attn_v$v.txt 2>&1; done
cat gpurun_out/r3_pmc_attn_v8.txt
