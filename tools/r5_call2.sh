#!/bin/bash
# r5 GPU call 2: the whole GPU suite after the ABI 9 / prune / renderer changes, host launch cost, lanes with start skew
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -rf --no-header 2>&1 | tail -60 > gpurun_out/r5_c2_pytest.log
for b in 16 8 4; do timeout 300 python tools/host_launch_cost.py $b 2>/dev/null | tail -1; done > gpurun_out/r5_c2_host.log
ROUNDS=1 timeout 900 bash tools/r4_ab_pipeline.sh base:LN3D_LANES=1 xcd0:LN3D_LANES=2:LN3D_LANE_MASK=xcd:LN3D_LANE_SKEW_US=0 xcd150:LN3D_LANES=2:LN3D_LANE_MASK=xcd:LN3D_LANE_SKEW_US=150 xcd300:LN3D_LANES=2:LN3D_LANE_MASK=xcd:LN3D_LANE_SKEW_US=300 cu150:LN3D_LANES=2:LN3D_LANE_MASK=cu:LN3D_LANE_SKEW_US=150 x4:LN3D_LANES=4:LN3D_LANE_MASK=xcd:LN3D_LANE_SKEW_US=110 > /dev/null 2>&1
cp gpurun_out/r4_ab_pipeline3.log gpurun_out/r5_c2_lanes.log
cat gpurun_out/r5_c2_pytest.log gpurun_out/r5_c2_host.log gpurun_out/r5_c2_lanes.log
