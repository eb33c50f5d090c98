cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3_pytest13.log 2>&1; tail -4 gpurun_out/r3_pytest13.log
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_c.json 2> gpurun_out/r3_bench_c.err; cut -c1-330 gpurun_out/r3_bench_c.json
