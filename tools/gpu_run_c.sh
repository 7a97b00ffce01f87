set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(cd /tmp && TMPDIR=/tmp rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r3_counters_avail.txt 2>&1 || true)
export PMC_BENCH_ARGS="--no-cpu-baseline --no-dense-mfma --steps 1 --warmup 1"
bash tools/pmc_variant.sh base "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" > gpurun_out/r3_pmc_decode_lean.txt 2>&1
cat gpurun_out/r3_pmc_decode_lean.txt
