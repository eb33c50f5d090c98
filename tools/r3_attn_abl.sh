#!/bin/bash
# builds (in the container) the attn_bench variants: instruction order x ablation bits of attn_kres_kernel
set -e
mkdir -p build
for o in 0 1 2; do for a in 0 1 2 4 8 16 32 3 12 28; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -DLN3D_KRES_ORDER=$o -DLN3D_KRES_ABL=$a tools/attn_bench.hip -o build/attn_bench_o${o}_a${a} &
done; wait; done
