#!/bin/bash
# Reproducer of profiles/r6_render_opsel.md: csrc/render.hip built WITHOUT the even-register copies of the odd tap weights (hipcc then broadcasts
# them into the packed FMAs with op_sel = 1: tools/check_isa.py lists the instructions), linked with the shipped objects into ab/libln3d_opsel_repro.so.
# On the GPU box:  LN3D_LIB=ab/libln3d_opsel_repro.so python tools/render_repeat_diff.py   -> tens of thousands of differing values per launch pair
#                  python tools/render_repeat_diff.py                                       -> identical
# and with -DRENDER_LDS_PAD=90112 added (one wave per SIMD) the repro build is identical too.
set -e
cd "$(dirname "$0")/.."
python __graft_entry__.py > /dev/null
mkdir -p ab
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC"
$H -DLN3D_RENDER_OPSEL_REPRO $EXTRA -c ln3diff_amd/csrc/render.hip -o build/render_opsel_repro.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libln3d_opsel_repro.so build/gemm_bf16.o build/attention.o build/dit_ops.o build/render_opsel_repro.o \
  build/conv_ops.o build/mesh.o build/runtime.o build/unet_ops.o
python tools/check_isa.py ab/libln3d_opsel_repro.so | head -4 || true
