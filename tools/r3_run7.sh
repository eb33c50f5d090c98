#!/bin/bash
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_entry_gpu.py tests/test_i23d_gpu.py -q -s > gpurun_out/r3_pytest7.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|FAILED|fold vs|dopri5 fp32|Error" gpurun_out/r3_pytest7.log | tail -30
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err; echo "bench rc $?"; cut -c1-230 gpurun_out/r3_bench_b.json; grep -o '"golden_check": {[^}]*}' gpurun_out/r3_bench_b.json
LN3D_NO_UC_FOLD=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r3_bench_b_nofold.json 2>/dev/null; cut -c1-230 gpurun_out/r3_bench_b_nofold.json
