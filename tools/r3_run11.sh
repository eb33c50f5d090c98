# renderer with / without hipcc's SLP vectoriser (packed fp32): timing, determinism, goldens; plus the pending I23D tests
cd $GRAFT_REPO_ROOT
echo "== library as built (-fno-slp-vectorize)"; timeout 200 python tools/render_bench.py 2>&1 | tail -4
timeout 600 python -m pytest tests/test_i23d_gpu.py tests/test_decode_gpu.py -x -q > gpurun_out/r3_pytest11a.log 2>&1; tail -3 gpurun_out/r3_pytest11a.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c ln3diff_amd/csrc/render.hip -o build/render.o 2>&1 | grep -v warning | head -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ln3diff_amd/libln3d_hip.so build/gemm_bf16.o build/attention.o build/dit_ops.o build/render.o build/conv_ops.o build/mesh.o
echo "== render.hip with SLP"; timeout 200 python tools/render_bench.py 2>&1 | tail -4
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_seams_gpu.py tests/test_geometry_gpu.py -x -q -k "render or rays or 512" > gpurun_out/r3_pytest11b.log 2>&1; tail -4 gpurun_out/r3_pytest11b.log
