# one GPU's share of configs[3] / configs[4] and r1's 8-view 128^2 configuration with the final build (profiles/r3_bench_*.json)
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py --arch DiT-XL/2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_cfg3.json 2>/dev/null; cut -c1-200 gpurun_out/r3_bench_cfg3.json
timeout 400 python bench.py --workload i23d --batch 2 --res 512 --views 24 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_cfg4.json 2>/dev/null; cut -c1-200 gpurun_out/r3_bench_cfg4.json
timeout 400 python bench.py --views 8 --res 128 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_bench_t23d_r1cfg.json 2>/dev/null; cut -c1-200 gpurun_out/r3_bench_t23d_r1cfg.json
