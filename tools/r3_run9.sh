cd $GRAFT_REPO_ROOT
timeout 120 ./build/l2_stream_bench > gpurun_out/r3_l2_stream_bench.txt 2>&1; cat gpurun_out/r3_l2_stream_bench.txt
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_i23d_gpu.py tests/test_entry_gpu.py -x -q > gpurun_out/r3_pytest9.log 2>&1; tail -5 gpurun_out/r3_pytest9.log; grep -h "vs bf16-operand\|plain DiT_I23D" gpurun_out/r3_pytest9.log
timeout 600 python -m pytest tests/test_decode_gpu.py -q -s -k bf16_operand 2>&1 | grep "bf16-operand" 
