#!/bin/bash
# same-box A/B of the whole bench line: the round-3 tree (git worktree of 551c014 under build/r3tree, its own library) against this
# tree, alternating, both workloads.  -> gpurun_out/r4_vs_r3.log
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r4_vs_r3.log
: > $L
one() { python $1/bench.py $2 --steps 2 --warmup 1 --no-cpu-baseline --no-probes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['golden_check'].get('rel_l2'))"; }
for r in 1 2; do
  for wl in "" "--workload i23d"; do
    echo "round $r [${wl:-t23d}] r3: $(one build/r3tree "$wl")" >> $L
    echo "round $r [${wl:-t23d}] r4: $(one . "$wl")" >> $L
  done
done
cat $L
