cd $GRAFT_REPO_ROOT
cp ln3diff_amd/libln3d_hip.so build/lib_cur.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -I ln3diff_amd/csrc -I include -c build/dit_ops_old.hip -o build/dit_ops_old.o 2>&1 | grep -v warning | head -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib_oldnorm.so build/gemm_bf16.o build/attention.o build/dit_ops_old.o build/render.o build/conv_ops.o build/mesh.o
for rep in 1 2; do
echo "== old norm kernel"; LN3D_LIB=build/lib_oldnorm.so timeout 100 python tools/norm_bench.py 2>&1 | tail -2
echo "== hoisted modulation loads"; timeout 100 python tools/norm_bench.py 2>&1 | tail -2
done
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_kernels_gpu.py tests/test_i23d_gpu.py -x -q > gpurun_out/r3_pytest18.log 2>&1; tail -3 gpurun_out/r3_pytest18.log
