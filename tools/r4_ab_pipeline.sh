#!/bin/bash
# same-box A/B of the whole bench line over library builds / switches: "tag:ENV=..:ENV=.." arguments
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r4_ab_pipeline3.log
: > $L
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for spec in "$@"; do
    tag=${spec%%:*}; envs=$(echo "${spec#*:}" | tr ':' ' ')
    v=$(env $envs python tools/bench_with_lib.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['golden_check'].get('rel_l2'))")
    echo "round $r $tag: $v" >> $L
  done
done
cat $L
