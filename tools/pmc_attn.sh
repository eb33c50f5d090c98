# PMC passes over the stand-alone attention bench (GPU box): one counter set per pass, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_TAG:-pmc_attn}
export ATTN_BENCH_CASES=1 ATTN_BENCH_VAR=${ATTN_BENCH_VAR:-1}
run() { n=$1; shift; timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $R/build/attn_bench > /dev/null 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
run sq3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE SQ_IFETCH
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "naive" in k: continue
            print(d.split("/")[-1], k)
            for c, v in cs.items():
                print("   %-28s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))
PY
