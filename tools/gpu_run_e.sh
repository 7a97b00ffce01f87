set -e
cd $GRAFT_REPO_ROOT
SRGPU_LIB=$GRAFT_REPO_ROOT/speechrecognition_amd/csrc/build/variants/libsrgpu_stamps.so timeout -k 10 300 python tools/decode_stamps_r3.py > gpurun_out/r3_decode_stamps2.txt 2>&1 || { tail -20 gpurun_out/r3_decode_stamps2.txt; exit 1; }
cat gpurun_out/r3_decode_stamps2.txt
export PMC_BENCH_ARGS="--no-cpu-baseline --no-dense-mfma --steps 1 --warmup 1"
bash tools/pmc_variant.sh base "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" 2>&1 | grep decode_fast
