#!/bin/bash
# PMC passes over tools/gemm_bench.hip (build/gemm_bench): clock (GRBM_GUI_ACTIVE / duration), matrix-pipe busy, SQ wait split,
# fabric bytes.  One counter set per pass, --kernel-trace only (the node pool refuses --pmc with the other trace domains).
#   CASE="fc1 GELU_ERF" bash tools/pmc_gemm.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_TAG:-pmc_gemm}
rm -rf $OUT; mkdir -p $OUT
BIN=${BIN:-$R/build/gemm_bench}
run() { n=$1; shift; timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $BIN 1 "${CASE:-fc1 GELU_ERF}" > $OUT/$n.log 2>&1; }
run sq1 GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run mem1 FETCH_SIZE
run mem2 WRITE_SIZE
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*/")):
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "gemm" not in k: continue
            dd = dur.get(k, [0])
            print(d.rstrip("/").split("/")[-1], k, "launches", len(dd), "mean duration %.1f us" % (sum(dd) / max(1, len(dd))))
            for c, v in cs.items():
                print("   %-28s n=%3d mean %.5g" % (c, len(v), sum(v) / len(v)))
PY
