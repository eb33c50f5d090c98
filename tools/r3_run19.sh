cd $GRAFT_REPO_ROOT
for rpw in 2 4; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -DNORM_RPW=$rpw -c ln3diff_amd/csrc/dit_ops.hip -o build/dit_ops_rpw$rpw.o 2>&1 | grep -v warning | head -3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib_rpw$rpw.so build/gemm_bf16.o build/attention.o build/dit_ops_rpw$rpw.o build/render.o build/conv_ops.o build/mesh.o
done
for rep in 1 2; do
echo "== 1 row per wave"; timeout 100 python tools/norm_bench.py 2>&1 | tail -2
echo "== 2 rows per wave"; LN3D_LIB=build/lib_rpw2.so timeout 100 python tools/norm_bench.py 2>&1 | tail -2
echo "== 4 rows per wave"; LN3D_LIB=build/lib_rpw4.so timeout 100 python tools/norm_bench.py 2>&1 | tail -2
done
