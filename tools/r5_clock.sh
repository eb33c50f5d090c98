#!/bin/bash
# r5: the clock every kernel of the denoise loop actually runs at, IN SITU: rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE over one bench
# step; per dispatch clock = (counter / 8 XCDs) / (End - Start).  (Counter collection serialises dispatches, so the step is slower than
# un-profiled - the clocks are what is read, not the times.)  -> gpurun_out/r5_clock.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_clk
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_clk -- python $R/bench.py --steps 1 --warmup 0 --no-probes --no-cpu-baseline --unfolded-steps 0 --sample-steps 40 > $R/gpurun_out/r5_clock_bench.json 2> $R/gpurun_out/r5_clock.err
python3 - <<PY > $R/gpurun_out/r5_clock.md
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob('/tmp/prof_clk/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE':
            continue
        dur = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
        if dur <= 0:
            continue
        a = acc[r['Kernel_Name'][:70]]
        a[0] += 1; a[1] += float(r['Counter_Value']) / 8.0; a[2] += dur
print('# r5 - in-situ clock per kernel of one configs[1] step (40 of the 250 EulerEDM steps; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE)')
print('| kernel | dispatches | avg us (profiled) | clock GHz = GRBM_GUI_ACTIVE / 8 / duration |')
print('|---|---|---|---|')
for k, (n, cyc, ns) in sorted(acc.items(), key=lambda kv: -kv[1][2])[:14]:
    print('| \`%s\` | %d | %.1f | %.2f |' % (k, n, ns / n / 1e3, cyc / ns))
PY
cat $R/gpurun_out/r5_clock.md
