#!/bin/bash
# r5 GPU call 1: sampler seam tests, lanes A/B of the whole bench line, ray-marcher PMC passes
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_seams_gpu.py -x -q -m gpu -k "edm_sampler" 2>&1 | tail -15 > gpurun_out/r5_c1_pytest.log
ROUNDS=2 timeout 900 bash tools/r4_ab_pipeline.sh base:LN3D_LANES=1 none:LN3D_LANES=2:LN3D_LANE_MASK=none cu:LN3D_LANES=2:LN3D_LANE_MASK=cu xcd:LN3D_LANES=2:LN3D_LANE_MASK=xcd > /dev/null 2>&1
cp gpurun_out/r4_ab_pipeline3.log gpurun_out/r5_c1_lanes.log
timeout 600 bash tools/pmc_render.sh > gpurun_out/r5_c1_render_pmc.log 2>&1
cat gpurun_out/r5_c1_pytest.log gpurun_out/r5_c1_lanes.log
tail -60 gpurun_out/r5_c1_render_pmc.log
