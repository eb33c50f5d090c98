#!/bin/bash
# r6: same-box A/B of the WHOLE bench line, the kernel library as it was at the start of this session (ab/libln3d_r6start.so = csrc of b0b3666:
# r5 ray-marcher, 128-wide XL/2 heads are a host-side choice and do not enter configs[1] / [2] / [4]) against the in-tree library; 2 alternations
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_session_ab.log; : > $L
run() {  # tag, lib, bench args
  v=$(LN3D_LIB=$2 timeout 600 python tools/bench_with_lib.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['golden_check'].get('rel_l2'))")
  echo "$1: $v" >> $L
}
for r in 1 2; do
  run "round $r configs[1] start" ab/libln3d_r6start.so ""
  run "round $r configs[1] in-tree" "" ""
  run "round $r configs[2] start" ab/libln3d_r6start.so "--workload i23d"
  run "round $r configs[2] in-tree" "" "--workload i23d"
  run "round $r configs[4]-share start" ab/libln3d_r6start.so "--workload i23d --batch 2 --res 512"
  run "round $r configs[4]-share in-tree" "" "--workload i23d --batch 2 --res 512"
done
cat $L
