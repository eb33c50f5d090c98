# same-box A/B of the renderer: round-2 render.hip vs the current one (hand-packed interpolation FMAs, planar merge arrays), seeded scene
cd $GRAFT_REPO_ROOT
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ln3diff_amd/libln3d_hip.so build/gemm_bf16.o build/attention.o build/dit_ops.o $1 build/conv_ops.o build/mesh.o; }
cp ln3diff_amd/libln3d_hip.so build/lib_cur.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -I ln3diff_amd/csrc -I include -c build/render_r2.hip -o build/render_r2.o 2>&1 | grep -v warning | head -3
for rep in 1 2; do
  link build/render_r2.o; echo "== r2 render.hip"; timeout 200 python tools/render_bench.py 2>&1 | tail -3
  cp build/lib_cur.so ln3diff_amd/libln3d_hip.so; echo "== current"; timeout 200 python tools/render_bench.py 2>&1 | tail -3
done
