#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_samplers_gpu.py tests/test_entry_gpu.py -x -q -m gpu -s > gpurun_out/r6_call12_pytest.log 2>&1
tail -4 gpurun_out/r6_call12_pytest.log; grep -E "unet |edm churn" gpurun_out/r6_call12_pytest.log
bash tools/r6_slp.sh > /dev/null 2>&1; cat gpurun_out/r6_render_slp.log | cut -c1-400
for v in mfma small; do
  if [ $v = mfma ]; then S=bench.py; else S=tools/unet_bench_ab.py; fi
  timeout 600 python $S --workload unet --steps 1 --warmup 1 > gpurun_out/r6_bench_unet_$v.json 2> gpurun_out/r6_bench_unet_$v.err
  python -c "
import json,sys
r=[json.loads(l) for l in open('gpurun_out/r6_bench_unet_$v.json') if l.startswith('{')]
print('$v', r[-1]['value'], r[-1]['ms_per_step']) if r else print('$v no line', open('gpurun_out/r6_bench_unet_$v.err').read()[-800:])
"
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/unetprof -o u -- python $OLDPWD/bench.py --workload unet --steps 1 --warmup 0 --sample-steps 25 > /dev/null 2>&1)
python - <<'PY'
import glob, csv
fs = glob.glob('/tmp/unetprof/**/*kernel_stats.csv', recursive=True)
for f in fs:
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    with open('gpurun_out/r6_unet_kernel_stats.md', 'w') as o:
        o.write('| kernel | calls | avg us | share |\n|---|---|---|---|\n')
        for r in rows[:25]:
            line = '| %s | %s | %.1f | %.1f %% |' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot)
            o.write(line + '\n'); print(line)
PY
