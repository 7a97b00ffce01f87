set -e
cd $GRAFT_REPO_ROOT
SRGPU_LIB=$GRAFT_REPO_ROOT/speechrecognition_amd/csrc/build/variants/libsrgpu_stamps.so timeout -k 10 300 python tools/decode_stamps_r3.py > gpurun_out/r3_decode_stamps4.txt 2>&1 || { tail -20 gpurun_out/r3_decode_stamps4.txt; exit 1; }
cat gpurun_out/r3_decode_stamps4.txt
