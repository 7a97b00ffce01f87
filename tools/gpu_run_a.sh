set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 tools/fp64_lds_width 20000 > gpurun_out/r3_fp64_lds_width.txt 2>&1
cat gpurun_out/r3_fp64_lds_width.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests_a.log 2>&1 || { tail -30 gpurun_out/r3_tests_a.log; exit 1; }
tail -3 gpurun_out/r3_tests_a.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.err
cat gpurun_out/r3_bench_a.json | cut -c1-600
