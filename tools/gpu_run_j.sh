set -e
cd $GRAFT_REPO_ROOT
SRGPU_LIB=$GRAFT_REPO_ROOT/speechrecognition_amd/csrc/build/variants/libsrgpu_ppk.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "prefilter or golden_scores or cfg3 or overflow or nan or finite or full_size" > gpurun_out/r3_tests_j.log 2>&1 || { tail -40 gpurun_out/r3_tests_j.log; exit 1; }
tail -2 gpurun_out/r3_tests_j.log
bash tools/ab_bench.sh base ppk 2>&1 | grep -v "^W2026\|^E2026"
