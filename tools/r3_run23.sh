# appended-token K/V cache of the I23D blocks: tests, configs[2] with and without it (same box)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_i23d_gpu.py tests/test_geometry_gpu.py tests/test_fullsize_gpu.py tests/test_entry_gpu.py -x -q -s > gpurun_out/r3_pytest23.log 2>&1; tail -3 gpurun_out/r3_pytest23.log; grep -h "appended K/V\|I23D B=32\|i23d PixArt-L/2" gpurun_out/r3_pytest23.log
timeout 300 python bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r3_bench_i23d_cache.json 2> gpurun_out/r3_bench_i23d_cache.err; cut -c1-230 gpurun_out/r3_bench_i23d_cache.json; tail -2 gpurun_out/r3_bench_i23d_cache.err
LN3D_NO_APPEND_CACHE=1 timeout 300 python bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r3_bench_i23d_nocache.json 2>/dev/null; cut -c1-230 gpurun_out/r3_bench_i23d_nocache.json
