#!/bin/bash
# end-of-round measurement set (GPU box): PMC traffic (-> profiles/r4_pmc.json), kernel stats of both bench commands, the bench lines
# of configs[1] / [2] with their CPU baselines, the unfolded (LN3D_NO_UC_FOLD=1) figures, configs[3] (XL/2) and the dopri5 line.
# Everything lands in gpurun_out/; copy what is quoted into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_traffic.sh > gpurun_out/r4_pmc.log 2>&1; tail -4 gpurun_out/r4_pmc.log
cp gpurun_out/r4_pmc.json profiles/r4_pmc.json
bash tools/r4_prof.sh i23d z > gpurun_out/r4_prof_final.log 2>&1; head -16 gpurun_out/r4_z_kernel_stats_t23d.md | tail -13
timeout 600 python bench.py > gpurun_out/r4_bench_t23d.json 2> gpurun_out/r4_bench_t23d.err; cut -c1-200 gpurun_out/r4_bench_t23d.json
timeout 600 python bench.py --workload i23d > gpurun_out/r4_bench_i23d.json 2> gpurun_out/r4_bench_i23d.err; cut -c1-200 gpurun_out/r4_bench_i23d.json
LN3D_NO_UC_FOLD=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r4_bench_t23d_nofold.json 2>/dev/null; cut -c1-160 gpurun_out/r4_bench_t23d_nofold.json
LN3D_NO_UC_FOLD=1 timeout 600 python bench.py --workload i23d --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r4_bench_i23d_nofold.json 2>/dev/null; cut -c1-160 gpurun_out/r4_bench_i23d_nofold.json
timeout 600 python bench.py --workload i23d --ode-method dopri5 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r4_bench_i23d_dopri5.json 2>/dev/null; cut -c1-160 gpurun_out/r4_bench_i23d_dopri5.json
bash tools/r4_prof_cfg3.sh z > gpurun_out/r4_prof_cfg3.log 2>&1; cut -c1-200 gpurun_out/r4_z_bench_cfg3.json
# one GPU's share of configs[4]: I23D, 2 samples per GPU, 24 views @ 512^2 (+ the mesh step is timed in tests/test_mesh_gpu.py)
timeout 600 python bench.py --workload i23d --batch 2 --res 512 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r4_bench_cfg4.json 2>/dev/null; cut -c1-200 gpurun_out/r4_bench_cfg4.json
