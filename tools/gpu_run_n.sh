set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests_n.log 2>&1 || { tail -40 gpurun_out/r3_tests_n.log; exit 1; }
tail -2 gpurun_out/r3_tests_n.log
bash tools/ab_bench.sh base 2>&1 | grep -v "^W2026\|^E2026"
