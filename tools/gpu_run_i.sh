set -e
cd $GRAFT_REPO_ROOT
bash tools/ab_bench.sh base g512x8 2>&1 | grep -v "^W2026\|^E2026"
