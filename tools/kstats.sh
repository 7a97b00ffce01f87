#!/bin/bash
# Per-kernel average durations of one bench.py run under rocprofv3 --kernel-trace --stats (top rows).
# usage: gpurun -- 'tools/kstats.sh TAG [bench args...]'
set -e -o pipefail
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-dense-mfma --no-boundary --steps 5 --warmup 2 $* > $out/log.txt 2>&1
cd $GRAFT_REPO_ROOT
python3 - $out/s_kernel_stats.csv <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 6:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
