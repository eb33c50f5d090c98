#!/bin/bash
# r5 end-of-round measurement set, second pass (after the CFG-twins block-0 dedup went in; kernel sources and therefore
# profiles/r5_pmc.json unchanged since tools/r5_final.sh).  Everything lands in gpurun_out/; tools/r5_collect.sh copies what the docs quote.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
# the relocated attention experiment still builds, runs and checks itself
ATTN_BENCH_CASES=2 timeout 120 build/attn_bench > gpurun_out/r5_attn_bench_final.log 2>&1; cat gpurun_out/r5_attn_bench_final.log
sed -e 's/r4_/r5_/g' -e 's/# r4 /# r5 /' tools/r4_prof.sh > /tmp/r5_prof.sh
bash /tmp/r5_prof.sh i23d z > gpurun_out/r5_prof_final.log 2>&1; head -16 gpurun_out/r5_z_kernel_stats_t23d.md | tail -13
timeout 900 python bench.py > gpurun_out/r5_bench_t23d.json 2> gpurun_out/r5_bench_t23d.err; cut -c1-220 gpurun_out/r5_bench_t23d.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_t23d_20steps.json 2> gpurun_out/r5_bench_t23d_20steps.err; cut -c1-220 gpurun_out/r5_bench_t23d_20steps.json
timeout 900 python bench.py --workload i23d > gpurun_out/r5_bench_i23d.json 2> gpurun_out/r5_bench_i23d.err; cut -c1-220 gpurun_out/r5_bench_i23d.json
timeout 900 python bench.py --arch DiT-XL/2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5_bench_cfg3.json 2>/dev/null; cut -c1-200 gpurun_out/r5_bench_cfg3.json
timeout 600 python bench.py --workload i23d --batch 2 --res 512 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r5_bench_cfg4.json 2>/dev/null; cut -c1-200 gpurun_out/r5_bench_cfg4.json
timeout 600 python bench.py --workload i23d --ode-method dopri5 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r5_bench_i23d_dopri5.json 2>/dev/null; cut -c1-160 gpurun_out/r5_bench_i23d_dopri5.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 1500 python -m pytest tests -q -m gpu -rf --no-header 2>&1 | tail -15 > gpurun_out/r5_pytest_gpu_final.log; cat gpurun_out/r5_pytest_gpu_final.log
