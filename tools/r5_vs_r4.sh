#!/bin/bash
# same-box A/B of the whole bench line: the round-4 tree (git archive of dd2380f under build/r4tree, its own library) against this
# tree, alternating, both workloads.  -> gpurun_out/r5_vs_r4.log
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r5_vs_r4.log
: > $L
one() { python $1/bench.py $2 --steps 2 --warmup 1 --no-cpu-baseline --no-probes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['golden_check'].get('rel_l2'))"; }
for r in 1 2; do
  for wl in "" "--workload i23d"; do
    echo "round $r [${wl:-t23d}] r4: $(one build/r4tree "$wl")" >> $L
    echo "round $r [${wl:-t23d}] r5: $(one . "$wl")" >> $L
  done
done
cat $L
