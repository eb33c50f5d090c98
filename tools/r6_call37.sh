#!/bin/bash
# r6: which ray-marcher configuration in situ?  Whole bench line, one box, two alternations: decoder under the gather at 4-wave workgroups / 3 waves per
# SIMD (il_w4o3), decoder behind the gather at the same occupancy (seq_w4o3), decoder behind the gather at 8-wave workgroups / 4 per SIMD (in-tree)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_insitu.log; : > $L
run() {
  v=$(LN3D_LIB=$2 timeout 600 python tools/bench_with_lib.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['golden_check'].get('rel_l2'))")
  echo "$1: $v" >> $L
}
for r in 1 2; do
  for w in "configs[1]:" "configs[2]:--workload i23d"; do
    tag=${w%%:*}; args=${w#*:}
    run "round $r $tag il_w4o3" ab/libln3d_il_w4o3.so "$args"
    run "round $r $tag seq_w4o3" ab/libln3d_seq_w4o3.so "$args"
    run "round $r $tag seq_w8o4 (in-tree)" "" "$args"
  done
done
cat $L
