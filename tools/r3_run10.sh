cd $GRAFT_REPO_ROOT
for v in noslp slp slp_wait0 slp_snop slp_pad; do echo "== $v"; timeout 60 ./build/det_$v 2>&1 | grep "rep\|lanes" | head -6; done > gpurun_out/r3_render_det.txt 2>&1
cat gpurun_out/r3_render_det.txt
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_i23d_gpu.py tests/test_entry_gpu.py -x -q -s > gpurun_out/r3_pytest10.log 2>&1; tail -4 gpurun_out/r3_pytest10.log; grep -h "bf16-operand\|plain DiT_I23D\|resblock\|attention block\|upsample + conv\|norm_out" gpurun_out/r3_pytest10.log
