#!/bin/bash
# r6 end-of-round measurement set (GPU box): counter passes first (installed as profiles/r6_pmc.json in the box's copy so that the bench
# lines of this call carry `traffic` for exactly the sources that ran), kernel stats of both bench commands, the bench lines of
# configs[1 - 4] + dopri5 + the U-Net line, smoke, the whole GPU suite.  tools/r6_collect.sh copies what the docs quote into profiles/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/r6_pmc.sh > gpurun_out/r6_pmc.log 2>&1; tail -8 gpurun_out/r6_pmc.log
cp gpurun_out/r6_pmc.json profiles/r6_pmc.json
sed -e 's/r4_/r6_/g' -e 's/# r4 /# r6 /' tools/r4_prof.sh > /tmp/r6_prof.sh
bash /tmp/r6_prof.sh i23d z > gpurun_out/r6_prof_final.log 2>&1; head -16 gpurun_out/r6_z_kernel_stats_t23d.md | tail -13
timeout 900 python bench.py > gpurun_out/r6_bench_t23d.json 2> gpurun_out/r6_bench_t23d.err; cut -c1-220 gpurun_out/r6_bench_t23d.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_t23d_20steps.json 2> gpurun_out/r6_bench_t23d_20steps.err; cut -c1-220 gpurun_out/r6_bench_t23d_20steps.json
timeout 900 python bench.py --workload i23d > gpurun_out/r6_bench_i23d.json 2> gpurun_out/r6_bench_i23d.err; cut -c1-220 gpurun_out/r6_bench_i23d.json
timeout 900 python bench.py --arch DiT-XL/2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r6_bench_cfg3.json 2>/dev/null; cut -c1-200 gpurun_out/r6_bench_cfg3.json
timeout 600 python bench.py --workload i23d --batch 2 --res 512 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r6_bench_cfg4.json 2>/dev/null; cut -c1-200 gpurun_out/r6_bench_cfg4.json
timeout 600 python bench.py --workload i23d --ode-method dopri5 --steps 2 --warmup 1 --no-cpu-baseline --no-probes > gpurun_out/r6_bench_i23d_dopri5.json 2>/dev/null; cut -c1-160 gpurun_out/r6_bench_i23d_dopri5.json
timeout 600 python bench.py --workload unet --steps 1 --warmup 1 > gpurun_out/r6_bench_unet.json 2>/dev/null; cut -c1-160 gpurun_out/r6_bench_unet.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -q -m gpu -rf --no-header 2>&1 | tail -15 > gpurun_out/r6_pytest_gpu_final.log; cat gpurun_out/r6_pytest_gpu_final.log
