set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r3 --no-dense-mfma > gpurun_out/r3_profile.log 2>&1 || { tail -30 gpurun_out/r3_profile.log; exit 1; }
tail -40 gpurun_out/r3_profile.log
