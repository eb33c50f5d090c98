# FETCH_SIZE / WRITE_SIZE / TCC hit-miss passes for the fc1 GEMM and the self-attention kernel (GPU box), one counter per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mem
run() { n=$1; k=$2; shift 2; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python $R/tools/one_attn.py $k > /dev/null 2>&1; }
run g_f gemm_gelu FETCH_SIZE
run g_w gemm_gelu WRITE_SIZE
run g_t gemm_gelu TCC_HIT_sum TCC_MISS_sum
run a_f attn FETCH_SIZE
run a_w attn WRITE_SIZE
run a_t attn TCC_HIT_sum TCC_MISS_sum
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "gemm_bf16_ring64" not in k and "attn" not in k: continue
            for c, v in cs.items():
                print("%-6s %-52s %-14s n=%2d mean %.5g" % (d.split("/")[-1], k, c, len(v), sum(v) / len(v)))
PY
