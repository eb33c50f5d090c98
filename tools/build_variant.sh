#!/bin/bash
# build/libln3d_<tag>.so = the shipped library with extra compile flags (bench-only macros), for same-box A/B runs through
# tools/bench_with_lib.py / tools/r4_ab_pipeline.sh (LN3D_LIB=build/libln3d_<tag>.so).  usage: tools/build_variant.sh <tag> [-DMACRO=..]...
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p build/v_$tag
objs=""
for f in gemm_bf16 attention dit_ops render conv_ops mesh runtime unet_ops; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC "$@" -c ln3diff_amd/csrc/$f.hip -o build/v_$tag/$f.o &
  objs="$objs build/v_$tag/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/libln3d_$tag.so $objs
rm -rf build/v_$tag
ls -la build/libln3d_$tag.so
