set -e
cd $GRAFT_REPO_ROOT
SRGPU_LIB=$GRAFT_REPO_ROOT/speechrecognition_amd/csrc/build/variants/libsrgpu_bgstamps.so timeout -k 10 300 python tools/bigram_stamps_r3.py > gpurun_out/r3_bigram_stamps.txt 2>&1 || { tail -20 gpurun_out/r3_bigram_stamps.txt; exit 1; }
cat gpurun_out/r3_bigram_stamps.txt
bash tools/gpu_run_l.sh
