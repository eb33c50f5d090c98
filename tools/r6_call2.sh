#!/bin/bash
# r6 GPU call 2: changed tests, the bench line with clock / sustained-peak fields, vendor kernel names (csv), per-tile overhead of the
# one-wave-per-SIMD tile (x13): timeline build + no-epilogue build.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_seams_gpu.py tests/test_abi_cpu.py tests/test_unet_gpu.py -x -q -m gpu -s > gpurun_out/r6_call2_pytest.log 2>&1
tail -5 gpurun_out/r6_call2_pytest.log; grep -E "full chain|edm seam" gpurun_out/r6_call2_pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/r6_bench_a.json 2> gpurun_out/r6_bench_a.err; tail -c 3000 gpurun_out/r6_bench_a.json; tail -3 gpurun_out/r6_bench_a.err
L=gpurun_out/r6_x13_overhead.log; : > $L
for t in base abl1 abl8; do echo "=== $t" >> $L; timeout 200 build/gemm_bench_$t 3 "x13" >> $L 2>&1; timeout 100 build/gemm_bench_$t 3 "fc1 plain" >> $L 2>&1; done
grep -E "===|fc1|qkv|timeline|square" $L
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vend -o vend -- python $OLDPWD/tools/gemm_vs_blas.py > $OLDPWD/gpurun_out/r6_vendor_run.log 2>&1)
python - <<'PY'
import glob, csv
fs = glob.glob('/tmp/vend/**/*kernel_stats.csv', recursive=True)
print(fs)
for f in fs:
    rows = list(csv.DictReader(open(f)))
    with open('gpurun_out/r6_vendor_kernels.md', 'w') as o:
        for r in rows[:40]:
            line = '| %s | %s | %s |' % (r.get('Name'), r.get('Calls'), r.get('AverageNs'))
            o.write(line + '\n'); print(line[:600])
PY
