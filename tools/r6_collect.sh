#!/bin/bash
# after tools/r6_final.sh ran on the GPU box: copy what DESIGN.md quotes from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
G=gpurun_out
cat $G/r6_z_kernel_stats_t23d.md > profiles/r6_kernel_stats.md
echo >> profiles/r6_kernel_stats.md
cat $G/r6_z_kernel_stats_i23d.md >> profiles/r6_kernel_stats.md
cp $G/r6_pmc.json profiles/r6_pmc.json
for n in t23d t23d_20steps i23d cfg3 cfg4 i23d_dopri5 unet; do cp $G/r6_bench_$n.json profiles/r6_bench_$n.json; done
cp $G/r6_z_prof_t23d_bench.json profiles/r6_prof_bench_t23d.json
cp $G/r6_z_prof_i23d_bench.json profiles/r6_prof_bench_i23d.json
cp $G/r6_pytest_gpu_final.log profiles/r6_pytest_gpu_final.log
python3 - <<'PY'
import json
for n in ['t23d','t23d_20steps','i23d','cfg3','cfg4','i23d_dopri5']:
    d = json.load(open('profiles/r6_bench_%s.json' % n))
    print('%-14s %8.4f samples/s  unfolded %s  %9.2f ms/step  golden %s  frac %s  raymarch %s' % (n, d['value'], d.get('value_unfolded'), d['ms_per_step'],
          d.get('golden_check', {}).get('rel_l2'), d.get('roofline', {}).get('frac'), (d.get('roofline_raymarch') or {}).get('frac')))
PY
