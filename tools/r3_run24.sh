# GEMM main loop: fragment reads issued two per MFMA slot from the start of a K substep (-DLN3D_GEMM_EARLY_RD) vs one per slot
cd $GRAFT_REPO_ROOT
echo "== shipped (one fragment read per MFMA slot)"; KBENCH_VENDOR=0 timeout 200 python tools/kbench.py 2>&1 | grep -i "fc1\|fc2\|qkv\|proj\|square\|to_q" | head -12
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -DLN3D_GEMM_EARLY_RD -c ln3diff_amd/csrc/gemm_bf16.hip -o build/gemm_early.o 2>&1 | grep -v warning | head -3
cp ln3diff_amd/libln3d_hip.so build/lib_cur.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ln3diff_amd/libln3d_hip.so build/gemm_early.o build/attention.o build/dit_ops.o build/render.o build/conv_ops.o build/mesh.o
echo "== early reads"; KBENCH_VENDOR=0 timeout 200 python tools/kbench.py 2>&1 | grep -i "fc1\|fc2\|qkv\|proj\|square\|to_q" | head -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k gemm 2>&1 | tail -2
cp build/lib_cur.so ln3diff_amd/libln3d_hip.so
