#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (both workloads), summaries written to gpurun_out/ for profiles/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_t /tmp/prof_i
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3_prof_t23d_bench.json 2> $R/gpurun_out/r3_prof_t23d.err
DB=$(find /tmp/prof_t -name "*.db" | head -1)
python $R/tools/prof_db_summary.py $DB "# r3 - rocprofv3 --kernel-trace --stats of \`python bench.py --steps 1 --warmup 1 --no-cpu-baseline\` (T23D configs[1]), 1x MI355X" 22 > $R/gpurun_out/r3_kernel_stats_t23d.md
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -- python $R/bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline --no-probes > $R/gpurun_out/r3_prof_i23d_bench.json 2> $R/gpurun_out/r3_prof_i23d.err
DB=$(find /tmp/prof_i -name "*.db" | head -1)
python $R/tools/prof_db_summary.py $DB "# same build, \`python bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline --no-probes\` (I23D configs[2]: network batch 64)" 16 > $R/gpurun_out/r3_kernel_stats_i23d.md
cat $R/gpurun_out/r3_kernel_stats_t23d.md; cat $R/gpurun_out/r3_kernel_stats_i23d.md; cut -c1-200 $R/gpurun_out/r3_prof_i23d_bench.json
