#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
L=gpurun_out/r6_render_generic.log; : > $L
timeout 300 python tools/render_bench.py 256 >> $L 2>&1
for pr in objv128 shapenet eg3d48; do RENDER_PRESET=$pr timeout 300 python tools/render_bench.py 256 >> $L 2>&1; done
grep -v amdgpu.ids $L
G=gpurun_out/r6_graph_ab.log; : > $G
for r in 1 2; do for v in nograph graph; do
  if [ $v = graph ]; then export LN3D_GRAPH=1; else unset LN3D_GRAPH; fi
  echo "$v $(timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes --unfolded-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['golden_check'].get('rel_l2'))")" >> $G
done; done
unset LN3D_GRAPH
cat $G
