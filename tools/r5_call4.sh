#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_entry_gpu.py -q -m gpu -rf --no-header -s -k "unet or entry_points_run" 2>&1 | tail -60 > gpurun_out/r5_c4_pytest.log
cat gpurun_out/r5_c4_pytest.log
