#!/bin/bash
# PMC counters of one bench step for a library variant (tools/build_variant.py), one rocprofv3 pass per quoted group.
# usage: gpurun -- 'bash tools/pmc_variant.sh VARIANT "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." ["..."]'   (VARIANT "base" = in-tree library)
v=$1; shift
if [ "$v" != base ]; then export SRGPU_LIB=$GRAFT_REPO_ROOT/speechrecognition_amd/csrc/build/variants/libsrgpu_$v.so; fi
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${v}_$i
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pmc -d $out -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py ${PMC_BENCH_ARGS:---no-cpu-baseline --no-boundary --steps 1 --warmup 1} > $out.log 2>&1 || { echo "pass $i FAILED"; tail -5 $out.log; exit 1; }
  python3 - $out/p_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-48:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if any(v > 1e6 for v in d.values()):
        print(k, {c: f"{v / n[(k, c)]:.4g}" for c, v in d.items()}, "launches", max(n[(k, c)] for c in d))
PY
done
