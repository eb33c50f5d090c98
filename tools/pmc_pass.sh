cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, kernel-arg, counters...
  n=$1; k=$2; shift 2
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_e/$n -- python $R/tools/one_attn.py $k > /dev/null 2>&1
}
run g_mem gemm_gelu FETCH_SIZE WRITE_SIZE
run g_sq1 gemm_gelu SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run g_sq2 gemm_gelu SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU
run g_sq3 gemm_gelu SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run a_mem attn FETCH_SIZE WRITE_SIZE
run a_sq2 attn SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_VALU
find $R/gpurun_out/pmc_e -name "*counter_collection.csv" | head -20
