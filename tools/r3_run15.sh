cd $GRAFT_REPO_ROOT
timeout 200 python tools/render_bench.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_vit_gpu.py tests/test_kernels_gpu.py tests/test_seams_gpu.py tests/test_geometry_gpu.py tests/test_fullsize_gpu.py tests/test_entry_gpu.py -x -q -k "render or rays or 512 or vit or image or embed or final or patch or seams or stub or determin or config4" > gpurun_out/r3_pytest15.log 2>&1; tail -4 gpurun_out/r3_pytest15.log; grep -h "preprocess" gpurun_out/r3_pytest15.log | head -8
