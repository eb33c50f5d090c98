#!/bin/bash
# r6: the default bench lines with the final bench.py (roofline_raymarch.l1_path_busy_est) on the final sources
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r6_bench_t23d.json 2> gpurun_out/r6_bench_t23d.err; cut -c1-200 gpurun_out/r6_bench_t23d.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_t23d_20steps.json 2> gpurun_out/r6_bench_t23d_20steps.err; cut -c1-200 gpurun_out/r6_bench_t23d_20steps.json
timeout 900 python bench.py --workload i23d > gpurun_out/r6_bench_i23d.json 2> gpurun_out/r6_bench_i23d.err; cut -c1-200 gpurun_out/r6_bench_i23d.json
python - <<'PY'
import json
for n in ('t23d', 't23d_20steps', 'i23d'):
    d = json.load(open('gpurun_out/r6_bench_%s.json' % n))
    r = d.get('roofline_raymarch') or {}
    print(n, d['value'], 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), '| raymarch frac', r.get('frac'), 'l1', r.get('l1_path_busy_est'), 'ms/view', r.get('ms_per_view'), '| sha', d.get('bench_py_sha16'))
PY
