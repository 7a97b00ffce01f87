#!/bin/bash
# A/B timing of libsrgpu variants (tools/build_variant.py) on the GPU box: prints the per-kernel average durations
# of one bench run per variant.  usage: tools/ab_bench.sh NAME [NAME ...]   ("base" = the in-tree library; AB_ARGS="--config cfg5" etc. go to bench.py)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset SRGPU_LIB; else export SRGPU_LIB=$GRAFT_REPO_ROOT/speechrecognition_amd/csrc/build/variants/libsrgpu_$v.so; fi
  out=$GRAFT_REPO_ROOT/gpurun_out/ab_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --kernel ${AB_KERNEL:-prefilter} --no-cpu-baseline --no-dense-mfma --no-boundary --steps ${AB_STEPS:-2} --warmup 1 $AB_ARGS > $out.log 2>&1 || { echo "$v FAILED"; tail -5 $out.log; exit 1; }
  echo "== $v  step $(grep -o '"ms_per_step": [0-9.]*' $out.log | head -1)"
  python3 - $out/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 1e5:
        print(f'   {r["Name"][:60]:60s} {float(r["AverageNs"])/1e6:8.3f} ms')
PY
done
