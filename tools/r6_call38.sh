#!/bin/bash
# r6: after the last (comment-only) edit of render.hip - the counter passes again (entries are keyed by the source hash) and the two default
# bench lines on that exact source
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
bash tools/r6_pmc.sh > gpurun_out/r6_pmc.log 2>&1; tail -8 gpurun_out/r6_pmc.log
cp gpurun_out/r6_pmc.json profiles/r6_pmc.json
timeout 900 python bench.py > gpurun_out/r6_bench_t23d.json 2> gpurun_out/r6_bench_t23d.err; cut -c1-200 gpurun_out/r6_bench_t23d.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_t23d_20steps.json 2> gpurun_out/r6_bench_t23d_20steps.err; cut -c1-200 gpurun_out/r6_bench_t23d_20steps.json
timeout 900 python bench.py --workload i23d > gpurun_out/r6_bench_i23d.json 2> gpurun_out/r6_bench_i23d.err; cut -c1-200 gpurun_out/r6_bench_i23d.json
python - <<'PY'
import json
for n in ('t23d', 't23d_20steps', 'i23d'):
    d = json.load(open('gpurun_out/r6_bench_%s.json' % n))
    print(n, d['value'], 'traffic', d['roofline'].get('traffic'), 'raymarch traffic', (d.get('roofline_raymarch') or {}).get('traffic'), (d.get('roofline_raymarch') or {}).get('frac'))
PY
