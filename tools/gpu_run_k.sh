set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_feeder.py tests/test_gpu_configs.py tests/test_gpu_traceback.py -m gpu -x -q > gpurun_out/r3_tests_k.log 2>&1 || { tail -40 gpurun_out/r3_tests_k.log; exit 1; }
tail -2 gpurun_out/r3_tests_k.log
timeout -k 10 300 python tools/time_batch_boundary.py > gpurun_out/r3_batch_boundary.txt 2>&1 || { tail gpurun_out/r3_batch_boundary.txt; exit 1; }
cat gpurun_out/r3_batch_boundary.txt
