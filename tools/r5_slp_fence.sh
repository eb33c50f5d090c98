#!/bin/bash
# r5: does the RAY_FENCE contain the SLP irreproducibility?  build/libln3d_slpfence.so = the library with hipcc's SLP vectoriser ON
# (default -O3: 130 trees in r4) and the fence of csrc/render.hip in place; the repeat tests run against it.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
cp ln3diff_amd/libln3d_hip.so /tmp/keep.so
cp build/libln3d_slpfence.so ln3diff_amd/libln3d_hip.so
python -m pytest tests/test_render_gpu.py -q -k "256_properties or bitwise_repeatable or sweep_100" 2>&1 | tail -4 > gpurun_out/r5_slp_fence.log
cp /tmp/keep.so ln3diff_amd/libln3d_hip.so
cat gpurun_out/r5_slp_fence.log
