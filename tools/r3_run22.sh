# renderer: skip gather + decoder for rays that miss the box: same-box A/B against the previous build, goldens, determinism
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== previous build"; LN3D_LIB=build/lib_prev.so timeout 200 python tools/render_bench.py 2>&1 | tail -3
echo "== box-miss skip"; timeout 200 python tools/render_bench.py 2>&1 | tail -3
done
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_seams_gpu.py tests/test_geometry_gpu.py tests/test_fullsize_gpu.py tests/test_mesh_gpu.py tests/test_entry_gpu.py -x -q > gpurun_out/r3_pytest22.log 2>&1; tail -3 gpurun_out/r3_pytest22.log
