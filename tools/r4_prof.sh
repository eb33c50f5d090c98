#!/bin/bash
# bench line + rocprofv3 --kernel-trace --stats of the bench command (T23D; pass "i23d" as $1 for configs[2] too) -> gpurun_out/r4_*
TAG=${2:-a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r4_${TAG}_bench_t23d.json 2> $R/gpurun_out/r4_${TAG}_bench_t23d.err
cut -c1-400 $R/gpurun_out/r4_${TAG}_bench_t23d.json
rm -rf /tmp/prof_t /tmp/prof_i
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r4_${TAG}_prof_t23d_bench.json 2> $R/gpurun_out/r4_${TAG}_prof_t23d.err
DB=$(find /tmp/prof_t -name "*.db" | head -1)
python $R/tools/prof_db_summary.py $DB "# r4 ($TAG) - rocprofv3 --kernel-trace --stats of \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline\` (T23D configs[1]), 1x MI355X" 24 > $R/gpurun_out/r4_${TAG}_kernel_stats_t23d.md
cat $R/gpurun_out/r4_${TAG}_kernel_stats_t23d.md
if [ "$1" == "i23d" ]; then
  python $R/bench.py --workload i23d --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r4_${TAG}_bench_i23d.json 2> $R/gpurun_out/r4_${TAG}_bench_i23d.err
  cut -c1-400 $R/gpurun_out/r4_${TAG}_bench_i23d.json
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -- python $R/bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline --no-probes > $R/gpurun_out/r4_${TAG}_prof_i23d_bench.json 2> $R/gpurun_out/r4_${TAG}_prof_i23d.err
  DB=$(find /tmp/prof_i -name "*.db" | head -1)
  python $R/tools/prof_db_summary.py $DB "# same build, \`python bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline --no-probes\` (I23D configs[2]: network batch 64)" 18 > $R/gpurun_out/r4_${TAG}_kernel_stats_i23d.md
  cat $R/gpurun_out/r4_${TAG}_kernel_stats_i23d.md
fi
