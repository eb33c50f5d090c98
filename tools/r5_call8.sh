#!/bin/bash
# r5 GPU call 8: ablations of attn_kres1w_kernel (LN3D_K1W_ABL bits; wrong results by construction, timing only) + issue counters
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r5_attn1w_abl.log; : > $L
for a in 0 1 2 4 8 16 32 34 64 65 28 128 98 127; do
  echo "== ABL $a" >> $L
  ATTN_BENCH_CASES=1 ATTN_BENCH_VAR=1 timeout 60 build/attn1w_a$a 2>&1 | grep kres1w >> $L
done
cat $L
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  ATTN_BENCH_CASES=1 ATTN_BENCH_VAR=1 timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/r5_a1w_pmc_$tag -o p -f csv -- $R/build/attn1w_a0 > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/r5_a1w_pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'kres1w' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
with open('gpurun_out/r5_attn1w_pmc.txt', 'w') as o:
    for k, (v, n) in sorted(acc.items()):
        line = '%-28s mean per launch %.4g  (%d rows)' % (k, v / max(n, 1), n)
        print(line); o.write(line + '\n')
PY
