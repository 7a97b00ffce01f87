set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_traceback.py tests/test_gpu_feeder.py tests/test_gpu_configs.py tests/test_real_speech.py tests/test_gpu_big_lexicon.py -m gpu -x -q > gpurun_out/r3_tests_g.log 2>&1 || { tail -40 gpurun_out/r3_tests_g.log; exit 1; }
tail -2 gpurun_out/r3_tests_g.log
SRGPU_LIB=$GRAFT_REPO_ROOT/speechrecognition_amd/csrc/build/variants/libsrgpu_stamps.so timeout -k 10 300 python tools/decode_stamps_r3.py > gpurun_out/r3_decode_stamps3.txt 2>&1 || { tail -20 gpurun_out/r3_decode_stamps3.txt; exit 1; }
cat gpurun_out/r3_decode_stamps3.txt
bash tools/ab_bench.sh base 2>&1 | grep -v "^W2026\|^E2026"
