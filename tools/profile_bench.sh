#!/bin/bash
# Profiles `python3 bench.py` on the GPU box: kernel-trace statistics, then PMC counters in separate passes (HBM
# traffic, clocks, pipe activity), as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# a --pmc pass carries --kernel-trace only, never --sys-trace / --runtime-trace or the hip/hsa/memory-copy domains).  Raw CSVs land in gpurun_out/profile_<tag>/; summarise with
# tools/summarize_profile.py and copy the summary into profiles/.
# usage: gpurun -- 'tools/profile_bench.sh TAG [bench args...]'
set -e -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/profile_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--no-cpu-baseline --no-boundary --steps 3 --warmup 1 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/stats.log 2>&1
grep '"metric"' $out/stats.log | tail -1 > $out/bench_under_profiler.json
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc -d $out/pmc$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-boundary --steps 1 --warmup 1 $* > $out/pmc$i.log 2>&1
  echo "pass $i ($pmc) done"
done
cd $GRAFT_REPO_ROOT && python3 tools/summarize_profile.py $out
