#!/bin/bash
# r6 GPU call 1: power / clock trace by phase, priority / start-skew variants of the two-workgroups-per-CU tile,
# the vendor GEMM's kernel names at the DiT shapes (study only).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/r6_power.py > gpurun_out/r6_power.md 2> gpurun_out/r6_power.err
tail -40 gpurun_out/r6_power.md
L=gpurun_out/r6_prio.log; : > $L
for r in 1 2; do for t in p0 p1 p2a p2b; do
  echo "=== round $r $t" >> $L
  timeout 120 build/gemm_bench_$t 3 "x14" >> $L 2>&1
  timeout 120 build/gemm_bench_$t 3 "half" >> $L 2>&1
done; done
grep -E "===|GATE_RES|CROSS|fc1|qkv" $L | head -120
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/vend -o vend -- python $OLDPWD/tools/gemm_vs_blas.py > $OLDPWD/gpurun_out/r6_vendor_run.log 2>&1)
python - <<'PY'
import glob, csv
fs = glob.glob('/tmp/vend/**/*kernel_stats.csv', recursive=True)
print(fs)
for f in fs:
    rows = list(csv.DictReader(open(f)))
    with open('gpurun_out/r6_vendor_kernels.md', 'w') as o:
        for r in rows[:40]:
            line = '| %s | %s | %s |' % (r.get('Name'), r.get('Calls'), r.get('AverageNs'))
            o.write(line + '\n'); print(line[:400])
PY
