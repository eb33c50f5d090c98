#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_geometry_gpu.py tests/test_i23d_gpu.py tests/test_entry_gpu.py -q -s > gpurun_out/r3_pytest_geom.log 2>&1; echo "pytest geom rc $?"
grep -E "passed|failed|FAILED|Error|XL/2|dopri5|grid192|^512 |assert" gpurun_out/r3_pytest_geom.log | tail -40
