#!/bin/bash
# same-box A/B of attention builds (tools/attn_bench.hip: self-checking against a naive fp32 kernel, bitwise repeat check)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r4_attn_ab.log
: > $L
for r in 1 2; do for b in "$@"; do echo "=== round $r $b" >> $L; ATTN_BENCH_CASES=${CASES:-5} timeout 300 build/$b >> $L 2>&1; done; done
cat $L
