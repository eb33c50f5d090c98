#!/bin/bash
# in-situ kernel stats of the T23D bench command under environment settings: "tag:ENV=..:ENV=.." arguments
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r4_prof_quick.log
: > $L
for spec in "$@"; do
  tag=${spec%%:*}; envs=$(echo "${spec#*:}" | tr ':' ' ')
  rm -rf /tmp/prof_$tag
  env $envs timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/tools/bench_with_lib.py --steps 1 --warmup 1 --no-cpu-baseline --no-probes $BENCH_ARGS > /tmp/b_$tag.json 2>/dev/null
  DB=$(find /tmp/prof_$tag -name "*.db" | head -1)
  echo "=== $tag: $(python -c "import json;d=json.load(open('/tmp/b_$tag.json'));print(d['value'],d['ms_per_step'])")" >> $L
  python $R/tools/prof_db_summary.py $DB "" ${ROWS:-12} | tail -${ROWS:-12} >> $L
done
cat $L
