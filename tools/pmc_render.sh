# PMC passes over the ray-marcher (GPU box): 4 views @ 256^2 per launch, one counter set per pass, --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_render
run() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- python $R/tools/render_bench.py 256 > /dev/null 2>&1; }
run mem1 FETCH_SIZE
run mem2 WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
run sq3 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "render_kernel" not in k: continue
            print(d.split("/")[-1], k)
            for c, v in cs.items():
                print("   %-28s n=%3d mean %.5g" % (c, len(v), sum(v) / len(v)))
PY
