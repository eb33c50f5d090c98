# renderer at 4 waves per SIMD (128 VGPRs, 26 spills): timing + tests, same box as the 3-wave build
cd $GRAFT_REPO_ROOT
cp ln3diff_amd/libln3d_hip.so build/lib_cur.so
echo "== RENDER_OCC 3 (shipped)"; timeout 200 python tools/render_bench.py 2>&1 | tail -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -DRENDER_OCC=4 -c ln3diff_amd/csrc/render.hip -o build/render_occ4.o 2>&1 | grep -v warning | head -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ln3diff_amd/libln3d_hip.so build/gemm_bf16.o build/attention.o build/dit_ops.o build/render_occ4.o build/conv_ops.o build/mesh.o
echo "== RENDER_OCC 4"; timeout 200 python tools/render_bench.py 2>&1 | tail -3
timeout 300 python -m pytest tests/test_render_gpu.py -x -q 2>&1 | tail -2
cp build/lib_cur.so ln3diff_amd/libln3d_hip.so
