# end-of-round measurement set: PMC traffic (-> profiles/r3_pmc.json), kernel stats of both bench commands, the two bench lines
# with their CPU baselines, the few test numbers quoted in DESIGN.md
cd $GRAFT_REPO_ROOT
bash tools/pmc_traffic.sh > gpurun_out/r3_pmc.log 2>&1; tail -6 gpurun_out/r3_pmc.log
cp gpurun_out/r3_pmc.json profiles/r3_pmc.json
bash tools/r3_prof.sh > gpurun_out/r3_prof_final.log 2>&1; head -14 gpurun_out/r3_kernel_stats_t23d.md | tail -11
timeout 600 python bench.py > gpurun_out/r3_bench_final_t23d.json 2> gpurun_out/r3_bench_final_t23d.err; cut -c1-260 gpurun_out/r3_bench_final_t23d.json
timeout 600 python bench.py --workload i23d > gpurun_out/r3_bench_final_i23d.json 2> gpurun_out/r3_bench_final_i23d.err; cut -c1-260 gpurun_out/r3_bench_final_i23d.json
timeout 300 python -m pytest tests/test_samplers_gpu.py tests/test_i23d_gpu.py -q -s -k "ddpm250 or plain or fold" 2>&1 | grep "DDPM-250\|plain DiT_I23D\|fold vs no fold"
