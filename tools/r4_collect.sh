#!/bin/bash
# after tools/r4_final.sh ran on the GPU box: copy what DESIGN.md quotes from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
G=gpurun_out
cat $G/r4_z_kernel_stats_t23d.md > profiles/r4_kernel_stats.md
echo >> profiles/r4_kernel_stats.md
cat $G/r4_z_kernel_stats_i23d.md >> profiles/r4_kernel_stats.md
cp $G/r4_z_kernel_stats_cfg3.md profiles/r4_kernel_stats_cfg3.md
cp $G/r4_pmc.json profiles/r4_pmc.json
for n in t23d i23d t23d_nofold i23d_nofold i23d_dopri5 cfg4; do cp $G/r4_bench_$n.json profiles/r4_bench_$n.json; done
cp $G/r4_z_bench_cfg3.json profiles/r4_bench_cfg3.json
cp $G/r4_z_prof_t23d_bench.json profiles/r4_prof_bench_t23d.json
cp $G/r4_z_prof_i23d_bench.json profiles/r4_prof_bench_i23d.json
python3 - <<'PY'
import json
for n in ['t23d','i23d','t23d_nofold','i23d_nofold','i23d_dopri5','cfg3','cfg4']:
    d = json.load(open('profiles/r4_bench_%s.json' % n))
    print('%-12s %8.4f samples/s  %9.2f ms/step  golden %s  frac %s' % (n, d['value'], d['ms_per_step'], d.get('golden_check', {}).get('rel_l2'), d.get('roofline', {}).get('frac')))
PY
