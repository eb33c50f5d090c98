#!/bin/bash
# one GPU's share of BASELINE configs[3] (DiT-XL/2, B = 8): bench line + rocprofv3 kernel stats -> gpurun_out/r6_${TAG}_*cfg3*
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python $R/bench.py --arch DiT-XL/2 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r6_${TAG}_bench_cfg3.json 2> $R/gpurun_out/r6_${TAG}_bench_cfg3.err
cut -c1-300 $R/gpurun_out/r6_${TAG}_bench_cfg3.json
rm -rf /tmp/prof_x
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -- python $R/bench.py --arch DiT-XL/2 --steps 1 --warmup 0 --no-cpu-baseline --no-probes > /dev/null 2> $R/gpurun_out/r6_${TAG}_prof_cfg3.err
DB=$(find /tmp/prof_x -name "*.db" | head -1)
python $R/tools/prof_db_summary.py $DB "# r6 ($TAG) - rocprofv3 --kernel-trace --stats of \`python bench.py --arch DiT-XL/2 --steps 1 --warmup 0 --no-cpu-baseline --no-probes\` (one GPU's share of configs[3]), 1x MI355X" 14 > $R/gpurun_out/r6_${TAG}_kernel_stats_cfg3.md
cat $R/gpurun_out/r6_${TAG}_kernel_stats_cfg3.md
