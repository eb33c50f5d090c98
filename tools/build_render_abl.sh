#!/bin/bash
# bench-only ablation builds of the library: render.hip with -DLN3D_RENDER_ABL=n (1 no decoder MLP, 2 no texel loads, 4 no compositing)
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py > /dev/null
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -DLN3D_RENDER_ABL=$n -c ln3diff_amd/csrc/render.hip -o build/render_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libln3d_abl$n.so build/gemm_bf16.o build/attention.o build/dit_ops.o build/render_abl$n.o build/conv_ops.o build/mesh.o
done
