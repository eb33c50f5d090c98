# I23D unconditional-branch fold: tests, entry points with the dopri5 default, configs[2] bench with and without the fold
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_i23d_gpu.py tests/test_seams_gpu.py tests/test_entry_gpu.py -x -q > gpurun_out/r3_pytest8.log 2>&1; tail -5 gpurun_out/r3_pytest8.log
timeout 300 python bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_i23d_fold.json 2> gpurun_out/r3_bench_i23d_fold.err; tail -c 600 gpurun_out/r3_bench_i23d_fold.json | cut -c1-300
LN3D_NO_UC_FOLD=1 timeout 300 python bench.py --workload i23d --steps 1 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r3_bench_i23d_nofold.json 2>/dev/null; cut -c1-250 gpurun_out/r3_bench_i23d_nofold.json
