# which compiler assumption makes the SLP build of render.hip produce different images?  (correctness hygiene: UB vs miscompile)
cd $GRAFT_REPO_ROOT
link() { /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ln3diff_amd/libln3d_hip.so build/gemm_bf16.o build/attention.o build/dit_ops.o build/render.o build/conv_ops.o build/mesh.o; }
try() { echo "== render.hip with: $*"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c ln3diff_amd/csrc/render.hip -o build/render.o 2>&1 | grep -v warning | head -3; link
  timeout 100 python tools/render_bench.py 256 2>&1 | tail -1
  timeout 300 python -m pytest tests/test_render_gpu.py -x -q 2>&1 | tail -2; }
try -fno-strict-aliasing
try -fno-slp-vectorize -fno-strict-aliasing
try -O1
try -mllvm -amdgpu-snop-padding=2
try -fno-slp-vectorize
timeout 300 python -m pytest tests/test_i23d_gpu.py -x -q -k plain 2>&1 | tail -2
