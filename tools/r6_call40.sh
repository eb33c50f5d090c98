#!/bin/bash
# r6: ring depth of the 80-wide attention kernel (3 shipped / 4 / 5) - isolated and on the whole XL/2 line (one GPU's share of configs[3])
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_attn80_nst.log; : > $L
for v in "" ab/libln3d_attn80_nst4.so ab/libln3d_attn80_nst5.so; do
  echo "== isolated ${v:-in-tree (3)}" >> $L
  LN3D_LIB=$v timeout 300 python tools/attn_general_ab.py 2>&1 | grep "72-in-80\|80-in-80 " >> $L
done
for r in 1 2; do
  for v in "" ab/libln3d_attn80_nst5.so; do
    x=$(LN3D_LIB=$v timeout 600 python tools/bench_with_lib.py --arch DiT-XL/2 --steps 2 --warmup 1 --no-cpu-baseline --no-probes 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['golden_check'].get('rel_l2'))")
    echo "round $r XL/2 ${v:-in-tree (3)}: $x" >> $L
  done
done
cat $L
