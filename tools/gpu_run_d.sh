set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_traceback.py tests/test_gpu_feeder.py -m gpu -x -q > gpurun_out/r3_tests_d.log 2>&1 || { tail -40 gpurun_out/r3_tests_d.log; exit 1; }
tail -2 gpurun_out/r3_tests_d.log
export AB_KERNEL=prefilter
bash tools/ab_bench.sh base samerow 2>&1 | grep -v "^W2026\|^E2026" | tee gpurun_out/r3_ab_d.txt
