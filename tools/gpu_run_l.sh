set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_bigram.py tests/test_gpu_configs.py -m gpu -x -q -k "bigram or cfg5" > gpurun_out/r3_tests_l.log 2>&1 || { tail -40 gpurun_out/r3_tests_l.log; exit 1; }
tail -2 gpurun_out/r3_tests_l.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_bigram_prof -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg5 --no-cpu-baseline --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/r3_bigram_bench.log 2>&1
grep -o '"ms_per_step": [0-9.]*' $GRAFT_REPO_ROOT/gpurun_out/r3_bigram_bench.log | head -1
python3 - $GRAFT_REPO_ROOT/gpurun_out/r3_bigram_prof/p_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 1e5:
        print(f'   {r["Name"][:60]:60s} {float(r["AverageNs"])/1e6:8.3f} ms x {r["Calls"]}')
PY
