#!/bin/bash
# r6: p4 kernel tests + same-box A/B of the whole bench line: shipped library (p4 picked for fc1) vs -DLN3D_P4_AUTO=0
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "persistent or x16" > gpurun_out/r6_p4_pytest.log 2>&1; tail -3 gpurun_out/r6_p4_pytest.log
L=gpurun_out/r6_p4_ab.log; : > $L
for r in 1 2; do
  for v in nop4 ship; do
    for wl in t23d i23d; do
      if [ $v = ship ]; then LIBARG=""; else LIBARG="build/libln3d_nop4.so"; fi
      echo "=== round $r $v $wl" >> $L
      LN3D_LIB=$LIBARG timeout 600 python tools/bench_with_lib.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --unfolded-steps 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['value'], r['ms_per_step'], 'fc1 in situ', r['roofline']['avg_us'], 'iso', r['roofline'].get('isolated_loop_avg_us'), r['roofline']['kernel'][:40], r['golden_check'].get('rel_l2'), r['golden_check'].get('rgb_rel_l2'))
" >> $L
    done
  done
done
cat $L
